#!/bin/bash
export TMPDIR=/tmp; R=$PWD; O=$R/gpurun_out/c35; mkdir -p $O
timeout 900 python -m pytest tests/test_conv_gpu.py -x -q 2>&1 | tail -3
cd /tmp
timeout 200 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $O/pmc_sq -o run -- python3 $R/scripts/profile_s2.py > $O/pmc3.log 2>&1
timeout 200 rocprofv3 --pmc GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_grbm -o run -- python3 $R/scripts/profile_s2.py > $O/pmc4.log 2>&1
cd $R
python3 scripts/summarize_profiles.py pmc $O/s2_pmc_summary.csv $(find $O/pmc_* -name "*counter_collection.csv")
rm -rf $O/pmc_*
grep "conv3x3_mfma" $O/s2_pmc_summary.csv
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "import json,sys; j=json.loads(sys.stdin.read()); print(j['value'], j['ms_per_step'])"
