import csv,glob,sys
f=glob.glob(sys.argv[1]+'/**/*kernel_trace.csv',recursive=True)[0]
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
t0=int(rows[0]['Start_Timestamp'])
# print last 100 dispatches: name, queue, start, end
for r in rows[-90:]:
    print(r['Kernel_Name'][:40].ljust(40), r.get('Queue_Id'), r.get('Stream_Id',''), int(r['Start_Timestamp'])-t0, int(r['End_Timestamp'])-t0, int(r['End_Timestamp'])-int(r['Start_Timestamp']))
