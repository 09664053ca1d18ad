"""Scratch: kernel-class totals of the gradient part from a rocprofv3 kernel trace of tools/grad_time.py."""
import sys, pandas as pd
df = pd.read_csv(sys.argv[1]).sort_values('Start_Timestamp').reset_index(drop=True)
df['us'] = (df.End_Timestamp - df.Start_Timestamp) / 1e3
g = df.index[df.Kernel_Name.str.contains('grad_kernel')].tolist()
k = df.index[df.Kernel_Name.str.contains('kmat_tile')].tolist()
# last evaluation: from the last kmat_tile before the last grad_kernel to that grad_kernel
end = g[-1]; start = max(i for i in k if i < end)
e = df.iloc[start:end + 1].copy()
red = e.index[e.Kernel_Name.str.contains('lml_reduce')].tolist()[0]
gp = e.loc[red + 1:]
gp['k'] = gp.Kernel_Name.str.replace(r'\(.*', '', regex=True).str.replace('void ', '')
print("gradient part: %.1f ms in %d launches; span %.1f ms" % (gp.us.sum() / 1e3, len(gp), (gp.End_Timestamp.max() - gp.Start_Timestamp.min()) / 1e6))
print(gp.groupby('k').us.agg(['size', 'sum', 'mean']).sort_values('sum', ascending=False).to_string())
