#!/bin/bash
export TMPDIR=/tmp; R=$PWD; O=$R/gpurun_out/c34; mkdir -p $O; cd /tmp
timeout 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $O/pmc_sq -o run -- python3 $R/scripts/profile_s2.py > $O/pmc3.log 2>&1
timeout 200 rocprofv3 --pmc GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_grbm -o run -- python3 $R/scripts/profile_s2.py > $O/pmc4.log 2>&1
cd $R
python3 scripts/summarize_profiles.py pmc $O/s2_pmc_summary.csv $(find $O/pmc_* -name "*counter_collection.csv")
rm -rf $O/pmc_*
grep "conv3x3_mfma" $O/s2_pmc_summary.csv
