import csv,glob,sys
f=sorted(glob.glob('/root/repo/gpurun_out/ll/runc/*kernel_trace.csv'))[-1]
rows=list(csv.DictReader(open(f)))
seq=[(r['Kernel_Name'].split('(')[0][-28:], (int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3) for r in rows]
idx=[i for i,(n,d) in enumerate(seq) if 'k_roipool_bins' in n]
for g in (idx[5], idx[16], idx[28]):
    print(' | '.join(f"{n.split('::')[-1][:18]} {d:.1f}" for n,d in seq[g:g+7]))
