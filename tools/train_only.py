"""The pre-training leg of bench.py alone (for rocprofv3: GPU-busy time per training step).  usage: train_only.py [steps]"""
import argparse, json, os, sys
os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")   # before torch: see gridmm_amd/train_graph.py
sys.path.insert(0, ".")
import torch
import bench
a = argparse.Namespace(batch=32)
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 6
print(json.dumps(bench.train_leg(a, torch.device("cuda:0"), steps=steps)))
