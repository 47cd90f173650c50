cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $R/gpurun_out/tr -- python3 $R/tools/bench_tgat.py --plain > /dev/null 2> $R/gpurun_out/tr.err
python3 - <<P
import csv,glob
f=glob.glob("$R/gpurun_out/tr/**/*kernel_trace.csv", recursive=True)[0]
rows=[(int(r["Start_Timestamp"]),int(r["End_Timestamp"]),r["Kernel_Name"][:50]) for r in csv.DictReader(open(f))]
rows.sort()
# last 2 fused calls: find the last 14*? kernels; simply take the last 120 kernels
tail=rows[-140:]
busy=sum(e-s for s,e,_ in tail); span=tail[-1][1]-tail[0][0]
print("last 140 kernels: span %.3f ms busy %.3f ms idle %.3f ms"%(span/1e6,busy/1e6,(span-busy)/1e6))
gaps=[(tail[i+1][0]-tail[i][1], tail[i][2], tail[i+1][2]) for i in range(len(tail)-1)]
gaps.sort(reverse=True)
for g in gaps[:12]: print("%.1f us after %s before %s"%(g[0]/1e3,g[1],g[2]))
P
rm -rf $R/gpurun_out/tr
