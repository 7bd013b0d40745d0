"""Per-block HBM rate of the Cexp-wide kernels of one training step, from a rocprofv3 kernel trace of `bench.py` (side stream off:
MUSCLE_WGRAD_STREAM=0, so durations are standalone).  usage: per_layer_hbm.py <kernel_trace.csv>"""
import csv, collections, os, re, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import muscle_amd

model = muscle_amd.MuSCLe(21, "efficientnet-b7", layers=3, last_pooling=False)
cfg = model.cfg
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
names = [r['Kernel_Name'] for r in rows]
st = [i for i, n in enumerate(names) if n.startswith('stem_im2col')]
ad = [i for i, n in enumerate(names) if n.startswith('adam_kernel')]
a = st[-1]
b = min(i for i in ad if i > a)
ph = rows[a:b]
dur = lambda r: (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
fp = [r for r in ph if 'colreduce_kernel<FPool' in r['Kernel_Name']]
sp = list(reversed([r for r in ph if r['Kernel_Name'].startswith('se_bn1_pool')]))
dwf = [r for r in ph if 'dw_fwd_kernel' in r['Kernel_Name']]
dwb = list(reversed([r for r in ph if 'dw_bwd_fused' in r['Kernel_Name'] or 'dw_bwd_data' in r['Kernel_Name']]))
N, h = 32, 224
print("B7 / 448x448 / batch 32, one step; us and TB/s of algorithmic bytes (d = depthwise output, e = depthwise input, both Cexp wide)")
print("blk  Cexp  Hin Hout k s | squeeze (1 d)  | se_bn1_pool (2 d) | dw forward (e + d) | dw backward (2 d + 2 e)")
tot, ideal = collections.Counter(), collections.Counter()
for i, blk in enumerate(cfg.blocks):
    ho = blk.out_size(h)
    d, e = N * ho * ho * blk.cexp * 4, N * h * h * blk.cexp * 4
    cells = []
    for nm, lst, by in (("squeeze", fp, d), ("se_bn1_pool", sp, 2 * d), ("dw_fwd", dwf, d + e), ("dw_bwd", dwb, 2 * d + 2 * e)):
        if i < len(lst):
            t = dur(lst[i]); cells.append("%7.1f %5.2f" % (t, by / t / 1e6)); tot[nm] += t; ideal[nm] += by / 5e6
        else:
            cells.append("      -     -")
    print("%3d %5d %4d %4d %d %d | %s  | %s     | %s      | %s" % (blk.index, blk.cexp, h, ho, blk.kernel, blk.stride, *cells))
    h = ho
print("totals, ms per step (measured / at 5 TB/s):", {k: (round(tot[k] / 1e3, 2), round(ideal[k] / 1e3, 2)) for k in tot})
