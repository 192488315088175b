import csv,glob,sys,collections
d=sys.argv[1]
cnt=collections.defaultdict(lambda: collections.defaultdict(list))
per=collections.defaultdict(float)
for f in glob.glob(d+'/**/*counter_collection.csv',recursive=True):
    for r in csv.DictReader(open(f)):
        per[(r['Kernel_Name'],r['Dispatch_Id'],r['Counter_Name'])]+=float(r['Counter_Value'])
for (k,i,c),v in per.items(): cnt[k][c].append(v)
dur=collections.defaultdict(list)
for f in glob.glob(d+'/**/*kernel_trace.csv',recursive=True):
    for r in csv.DictReader(open(f)):
        dur[r['Kernel_Name']].append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e6)
for k in cnt:
    if 'stream_gemm' not in k: continue
    ds=sorted(dur[k]); med=ds[len(ds)//2]
    m={c:sum(v)/len(v) for c,v in cnt[k].items()}
    gui=m.get('GRBM_GUI_ACTIVE',0); busy=m.get('SQ_VALU_MFMA_BUSY_CYCLES',0)
    print(k.split('(')[0][:70], 'n',len(ds),'med %.3f ms'%med, 'mfma_busy %.3f'%(busy/(gui*128) if gui else 0), 'clk %.3f GHz'%(gui/8/(sum(ds)/len(ds)*1e-3)/1e9), {c:round(v/1e6,1) for c,v in m.items()})
