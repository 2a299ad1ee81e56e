#!/bin/bash
R=$GRAFT_REPO_ROOT; export TMPDIR=/tmp; cd /tmp
O=$R/gpurun_out/cold_pmc; mkdir -p $O
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_EA0_WRREQ_sum TCC_EA0_RDREQ_sum" "GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $O/g$i -- python3 $R/tools/probe/cold_launch_pmc.py > $O/g$i.log 2>&1
done
python3 - $O > $O/summary.txt <<'PY'
import csv,glob,sys,collections
O=sys.argv[1]
for f in sorted(glob.glob(O+'/g*/*/*_counter_collection.csv')):
    rows=list(csv.DictReader(open(f)))
    disp=collections.OrderedDict()
    for r in rows:
        if 'step_kernel' in r['Kernel_Name']:
            disp.setdefault(int(r['Dispatch_Id']),{})[r['Counter_Name']]=float(r['Counter_Value'])
    ids=sorted(disp)[-10:]
    for k in ids: print(k, disp[k])
    print()
for f in sorted(glob.glob(O+'/g1.log')): print(open(f).read()[-600:])
PY
cat $O/summary.txt
