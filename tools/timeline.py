import csv,sys,glob
f=glob.glob(sys.argv[1]+'/**/*kernel_trace.csv',recursive=True)[0]
rows=list(csv.DictReader(open(f)))
ev=sorted((int(r["Start_Timestamp"]),int(r["End_Timestamp"]),r["Kernel_Name"][:60],r.get("Stream_Id",r.get("Queue_Id",""))) for r in rows)
blur=[e for e in ev if 'blur_direct' in e[2]]
print(len(ev),'kernels',len(blur),'blur launches')
b=blur[-12:]
t0=b[0][0]
for i in range(1,len(b)):
    print(f"blur {i}: start {(b[i][0]-t0)/1e3:9.1f} dur {(b[i][1]-b[i][0])/1e3:7.1f} gap-from-prev-end {(b[i][0]-b[i-1][1])/1e3:6.1f} period {(b[i][0]-b[i-1][0])/1e3:7.1f}")
lo=b[-4][0]
for e in ev:
    if e[0]>=lo: print(f"  {(e[0]-t0)/1e3:9.1f} +{(e[1]-e[0])/1e3:7.1f} q={e[3]} {e[2]}")
