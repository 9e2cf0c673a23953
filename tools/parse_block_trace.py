import csv,glob,sys
f=sorted(glob.glob('gpurun_out/k20/**/*kernel_trace.csv', recursive=True))[-1]
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
# find last k_mcmc_finish and walk back to its k_mcmc_begin
names=[r['Kernel_Name'] for r in rows]
idx=[i for i,n in enumerate(names) if n.startswith('k_mcmc_finish')]
for last in idx[-3:]:
    j=last
    while j>0 and not names[j].startswith('k_mcmc_begin'): j-=1
    blk=rows[j:last+1]
    t0=int(blk[0]['Start_Timestamp'])
    print('block of', len(blk), 'kernels; span', (int(blk[-1]['End_Timestamp'])-t0)/1e3, 'us')
    prev_end=None
    for r in blk[:4]+blk[-3:]:
        s,e=int(r['Start_Timestamp']),int(r['End_Timestamp'])
        print('   %-28s start %8.2f dur %6.2f gap %s' % (r['Kernel_Name'][:28], (s-t0)/1e3, (e-s)/1e3, '' if prev_end is None else round((s-prev_end)/1e3,2)))
        prev_end=e
