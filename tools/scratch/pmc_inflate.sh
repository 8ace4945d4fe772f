#!/bin/bash
export TMPDIR=/tmp
cd /tmp
for set in "SQ_INSTS_SALU SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_IFETCH SQ_INSTS_BRANCH SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE"; do
  rm -rf /tmp/pz; rocprofv3 --pmc $set -d /tmp/pz -o run --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/inflate_bench.py --groups 8192 --repeat 1 > /dev/null 2>&1
  python3 - <<'PY'
import csv,glob,collections
acc=collections.defaultdict(float)
for f in glob.glob('/tmp/pz/**/*counter_collection.csv',recursive=True):
    for r in csv.DictReader(open(f)):
        if 'inflate' in r['Kernel_Name']:
            acc[r['Counter_Name']]+=float(r['Counter_Value'])
print(dict(acc))
PY
done
