"""Scratch: replay the blocked recursion on the host to list every launch, zip it with a rocprofv3
kernel trace, and report achieved TFLOP/s per GEMM shape."""
import sys, re
import pandas as pd
T = 128
calls = []
def split(n): return ((n // T) // 2) * T
def potrf(n, m_extra=0):
    if n == T: calls.append(('base',)); return
    n1 = split(n); n2 = n - n1
    potrf(n1); trsm(n1, n2); calls.append(('gemm', 0, 1, n2, n2, n1)); potrf(n2)
def trsm(n, m):
    if n == T: calls.append(('gemm', 1, 0, m, T, T)); return
    n1 = split(n); n2 = n - n1
    trsm(n1, m); calls.append(('gemm', 0, 0, m, n2, n1)); trsm(n2, m)
N = int(sys.argv[2]) if len(sys.argv) > 2 else 32768
potrf(N)
df = pd.read_csv(sys.argv[1])
df['dur_us'] = (df['End_Timestamp'] - df['Start_Timestamp']) / 1e3
df = df.sort_values('Start_Timestamp').reset_index(drop=True)
idx = df.index[df['Kernel_Name'].str.contains('kmat_tile')].tolist()
e = df.iloc[idx[-1]:]
e = e[e['Kernel_Name'].str.contains('gemm_nt_f64|potrf_base')].reset_index(drop=True)
exp = [c for c in calls]
assert len(e) >= len(exp), (len(e), len(exp))
rows = []
for c, (_, r) in zip(exp, e.iterrows()):
    if c[0] == 'base':
        assert 'potrf_base' in r['Kernel_Name']; continue
    _, op, lower, M, Nn, K = c
    assert 'gemm' in r['Kernel_Name'], (c, r['Kernel_Name'])
    t128 = (M // T) * (M // T + 1) / 2 if lower else (M // T) * (Nn // T)
    fl = 2 * t128 * T * T * K
    m = re.match(r'void gemm_nt_f64_kernel<(\d+), (\d+)', r['Kernel_Name'])
    rows.append(dict(op=op, lower=lower, M=M, N=Nn, K=K, tile=m.group(1) + 'x' + m.group(2), flops=fl, us=r['dur_us']))
g = pd.DataFrame(rows)
agg = g.groupby(['op', 'lower', 'M', 'N', 'K', 'tile']).agg(n=('us', 'size'), us=('us', 'mean'), tot_ms=('us', lambda x: x.sum() / 1e3), flops=('flops', 'first')).reset_index()
agg['TF'] = agg['flops'] / agg['us'] / 1e6
agg['ideal_ms@67'] = agg['flops'] * agg['n'] / 67e12 * 1e3
agg['lost_ms'] = agg['tot_ms'] - agg['ideal_ms@67']
pd.set_option('display.width', 250)
print(agg.sort_values('lost_ms', ascending=False).head(40).to_string())
print("total gemm ms", agg['tot_ms'].sum(), "ideal", agg['ideal_ms@67'].sum())
