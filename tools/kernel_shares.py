import csv,sys,glob
for d in sys.argv[1:]:
    f=glob.glob(d+'/**/*kernel_stats.csv',recursive=True)
    if not f: print(d,'no stats'); continue
    rows=list(csv.DictReader(open(f[0])))
    al=[r for r in rows if 'alpine' in r['Name']]
    tot=sum(int(r['TotalDurationNs']) for r in al)
    print('==',d.split('/')[-1])
    for r in al[:9]:
        print('  ',r['Name'].split('(')[0].replace('void ','')[:64].ljust(66), r['Calls'].rjust(5), '%8.1f us'%(float(r['AverageNs'])/1e3), '%5.1f %%'%(100*int(r['TotalDurationNs'])/tot))
