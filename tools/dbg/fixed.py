import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tools"))
import bench_gemm as bg
bg.SHAPES = [(1824, 768, 64), (1824, 768, 128), (1824, 768, 256), (1824, 768, 768), (1824, 768, 1536), (1824, 768, 3072)]
bg.run([8, 108, 208, 15])
