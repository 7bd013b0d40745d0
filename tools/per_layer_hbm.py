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
# backward: stride-1 blocks run ONE fused kernel (3 reads + 1 write: dA, d, e in, gX out = 2 d + 2 e); stride-2 blocks run
# dw_bwd_weight (reads e and dd: e + d) and dw_bwd_data (reads dd, writes gX: d + e) behind a bn_bwd_apply pass of their own -
# round 3 priced the stride-2 rows' dw_bwd_data alone with the fused kernel's 2 d + 2 e (rows above 8 TB/s): each kernel now
# has its own byte model and its own column entry
dwb = list(reversed([r for r in ph if 'dw_bwd_fused' in r['Kernel_Name']]))
dwbw = list(reversed([r for r in ph if 'dw_bwd_weight' in r['Kernel_Name']]))
dwbd = list(reversed([r for r in ph if 'dw_bwd_data' in r['Kernel_Name']]))
N, h = 32, 224
print("B7 / 448x448 / batch 32, one step; us and TB/s of algorithmic bytes (d = depthwise output, e = depthwise input, both Cexp wide)")
print("blk  Cexp  Hin Hout k s | squeeze (1 d)  | se_bn1_pool (2 d) | dw forward (e + d) | dw backward: stride 1 fused (2 d + 2 e); stride 2 weight (e + d) + data (d + e)")
tot, ideal = collections.Counter(), collections.Counter()
i1 = i2 = 0
for i, blk in enumerate(cfg.blocks):
    ho = blk.out_size(h)
    d, e = N * ho * ho * blk.cexp * 4, N * h * h * blk.cexp * 4
    cells = []
    for nm, lst, by in (("squeeze", fp, d), ("se_bn1_pool", sp, 2 * d), ("dw_fwd", dwf, d + e)):
        if i < len(lst):
            t = dur(lst[i]); cells.append("%7.1f %5.2f" % (t, by / t / 1e6)); tot[nm] += t; ideal[nm] += by / 5e6
        else:
            cells.append("      -     -")
    if blk.stride == 1 and i1 < len(dwb):
        t = dur(dwb[i1]); i1 += 1
        cells.append("%7.1f %5.2f" % (t, (2 * d + 2 * e) / t / 1e6)); tot["dw_bwd"] += t; ideal["dw_bwd"] += (2 * d + 2 * e) / 5e6
    elif blk.stride == 2 and i2 < len(dwbw) and i2 < len(dwbd):
        tw, td = dur(dwbw[i2]), dur(dwbd[i2]); i2 += 1
        cells.append("weight %6.1f %5.2f, data %6.1f %5.2f" % (tw, (e + d) / tw / 1e6, td, (d + e) / td / 1e6))
        tot["dw_bwd_s2"] += tw + td; ideal["dw_bwd_s2"] += 2 * (d + e) / 5e6
    else:
        cells.append("      -     -")
    print("%3d %5d %4d %4d %d %d | %s  | %s     | %s      | %s" % (blk.index, blk.cexp, h, ho, blk.kernel, blk.stride, *cells))
    h = ho
print("totals, ms per step (measured / at 5 TB/s):", {k: (round(tot[k] / 1e3, 2), round(ideal[k] / 1e3, 2)) for k in tot})
