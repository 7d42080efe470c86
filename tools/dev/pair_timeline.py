import csv,glob,sys
f=glob.glob(sys.argv[1]+'/**/*kernel_trace.csv',recursive=True)[0]
rows=[(int(r["Start_Timestamp"]),int(r["End_Timestamp"]),r["Kernel_Name"].split("(")[0].replace("void ffhip::","").replace("ffhip::","")[:34],r["Queue_Id"],r.get("Stream_Id","?")) for r in csv.DictReader(open(f))]
rows.sort()
L=[r for r in rows if 'k_lstm_split' in r[2] or 'k_grumod_pack' in r[2] or 'k_lstm_pack' in r[2]]
gaps=[(L[i+1][0]-L[i][1])/1e3 for i in range(len(L)-1)]
big=[i for i,g in enumerate(gaps) if g>300]
idx=big[len(big)//2]
t0=L[idx][1]
print("layers end at 0; next layer starts at %.1f us; all gaps>300: %s"%(gaps[idx],[round(gaps[i]) for i in big]))
for r in rows:
    if r[1]>=L[idx][0] and r[0]<=L[idx+1][1]:
        print("%9.1f -> %9.1f  (%7.1f us) q%s s%s %s"%((r[0]-t0)/1e3,(r[1]-t0)/1e3,(r[1]-r[0])/1e3,r[3],r[4],r[2]))
