"""Summarise a rocprofv3 --kernel-trace --stats CSV into per-step categories.
usage: python tools/prof_summary.py <kernel_stats.csv> <steps_in_run>"""
import collections, csv, sys

def cat(name):
    if 'naive_conv' in name or 'miopenSp3' in name or 'igemm' in name or 'Im2d2Col' in name or 'Col2Im' in name: return 'MIOpen conv'
    if 'pairs_gemm_kernel' in name: return 'ftx spconv pairs_gemm'
    if 'spconv_reduce' in name: return 'ftx spconv reduce'
    if 'spconv_ostat' in name: return 'ftx spconv ostat (one-launch thin layers)'
    if 'pairs_wgrad_kernel' in name or 'wgrad_reduce' in name: return 'ftx spconv wgrad'
    if 'attn_' in name: return 'ftx attention'
    if name.startswith('Cijk'): return 'hipBLASLt GEMM (Cijk)'
    if 'bn_' in name: return 'ftx bn'
    if 'voxelize' in name or 'segment_reduce' in name or 'seg_prepare' in name: return 'ftx vox/devox (incl. sorted-segment reduces)'
    if 'add_ln_' in name or 'ln_params_final' in name or 'colsum_' in name: return 'ftx layernorm / column sums'
    if 'adam_kernel' in name: return 'ftx adam'
    if 'loss_' in name or name.startswith('sd_') or 'sd_' in name.split('(')[0]: return 'ftx loss / sample_down'
    if 'lift' in name or 'resample' in name: return 'ftx lift/resample'
    if any(k in name for k in ['hash', 'table_', 'kernel_map', 'count_kernel', 'iota', 'gather_coords', 'downsample', 'trilinear', 'floor_coords',
                               'fill_m1', 'koff_kernel', 'pairs_scatter', 'rocprim', 'sorted_rank']): return 'ftx index'
    if 'softmax' in name.lower(): return 'torch softmax'
    if 'elementwise' in name or 'FillFunctor' in name: return 'torch elementwise'
    if 'reduce_kernel' in name: return 'torch reduce'
    if 'layer_norm' in name.lower() or 'LayerNorm' in name or 'GammaBeta' in name: return 'torch layernorm'
    if 'batch_norm' in name.lower() or 'Batchnorm' in name or 'BatchNorm' in name: return 'torch/miopen batchnorm'
    if 'multi_tensor' in name or 'adam' in name.lower(): return 'optimizer'
    if 'copyBuffer' in name or 'fillBuffer' in name: return 'memcpy/memset'
    return 'other'

rows = list(csv.DictReader(open(sys.argv[1])))
steps = float(sys.argv[2])
agg = collections.OrderedDict()
for r in rows:
    d = agg.setdefault(cat(r['Name']), [0, 0.0])
    d[0] += int(r['Calls']); d[1] += float(r['TotalDurationNs']) / 1e6
tot = sum(v[1] for v in agg.values())
print('%-28s %12s %12s' % ('category', 'calls/step', 'ms/step'))
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print('%-28s %12.1f %12.3f' % (k, v[0] / steps, v[1] / steps))
print('%-28s %12.1f %12.3f' % ('TOTAL', sum(v[0] for v in agg.values()) / steps, tot / steps))
print('\ntop kernels:')
for r in rows[:int(sys.argv[3]) if len(sys.argv) > 3 else 25]:
    print('%6.2f%% %9.3f ms/step  calls/step %6.1f  avg %8.1f us  %s' % (float(r['Percentage']), float(r['TotalDurationNs']) / 1e6 / steps,
          int(r['Calls']) / steps, float(r['AverageNs']) / 1e3, r['Name'][:100]))
