import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
out=[]
for r in rows:
    n=r['Kernel_Name']
    if 'gemm' in n or 'Cijk' in n:
        out.append((n[:60], (int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3))
i=0
while i<len(out):
    j=i
    while j<len(out) and out[j][0]==out[i][0] and j-i<5: j+=1
    print(out[i][0], ' '.join(f"{x[1]:.1f}" for x in out[i:j]))
    i=j
