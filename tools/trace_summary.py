"""Per-kernel (name, grid) durations from a rocprofv3 --kernel-trace CSV; usage: trace_summary.py file.csv steps|auto [n]"""
import csv, collections, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
steps = sys.argv[2] if len(sys.argv) > 2 else '1'
# auto: a kernel that runs exactly once per step and nowhere else (the roofline leg re-launches the aggregation alone, and
# the aggregation is two kernels, so counting 'grid_aggregate' over-counts): cells_compact_kernel, else grid_project_kernel
def _count(name):
    return float(sum(name in r['Kernel_Name'] for r in rows))
steps = (_count('cells_embed_kernel') or _count('cells_compact_kernel') or _count('grid_project_kernel') or 1.0) if steps == 'auto' else float(steps)
top = int(sys.argv[3]) if len(sys.argv) > 3 else 30
def short(n):
    m = re.search(r'(linear_planes_kernel<[^>]*>|linear_kernel<[^>]*>|attention_rows_kernel<[^>]*>|attention_planes_kernel<\d>|attention_kernel|tokens_to_slab_kernel|transpose_v_kernel|grid_aggregate_pipe_kernel|grid_aggregate_kernel|layernorm_kernel<\d>|ln_dot_kernel|copy_rows_kernel|cells_compact_kernel|grid_bin_sort_kernel|grid_project_kernel|split_rows_kernel|fuse_logits|text_fragments|build_chunks|split_weight)', n)
    return m.group(1) if m else n[:48]
agg = collections.defaultdict(lambda: [0, 0.0])
for r in rows:
    k = (short(r['Kernel_Name']), r['Grid_Size_X'], r['Grid_Size_Y'], r['Grid_Size_Z'])
    agg[k][0] += 1
    agg[k][1] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:top]:
    print("%-50s grid=%s,%s,%s calls/step=%.1f avg=%.1fus per_step=%.1fus" % (k[0], k[1], k[2], k[3], v[0] / steps, v[1] / v[0], v[1] / steps))
print('steps counted', steps)
print('total per step us', sum(v[1] for v in agg.values()) / steps)
