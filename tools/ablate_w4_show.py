import csv,sys,glob
for d in sorted(glob.glob('/root/repo/gpurun_out/abl_w4_*_*/r_kernel_stats.csv')):
    tag=d.split('/')[-2]
    if sys.argv[1:] and not any(tag.endswith(a) for a in sys.argv[1:]): continue
    for r in csv.DictReader(open(d)):
        if 'conv_w4' in r['Name'] or 'conv_rb_kernel<64' in r['Name']: print(tag, r['Calls'], round(float(r['AverageNs'])/1000,1), r['Name'][30:90])
