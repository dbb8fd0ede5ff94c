#!/bin/bash
# refresh of the bf16 body kernels' PMC summary on the final code (four separate --pmc passes over scripts/profile_b16.py)
export TMPDIR=/tmp; R=$PWD; O=$R/gpurun_out/c62; mkdir -p $O
cd /tmp
timeout 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $O/pmc_sq -o run -- python3 $R/scripts/profile_b16.py > $O/pmc3.log 2>&1
timeout 200 rocprofv3 --pmc GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_grbm -o run -- python3 $R/scripts/profile_b16.py > $O/pmc4.log 2>&1
timeout 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o run -- python3 $R/scripts/profile_b16.py > $O/pmc1.log 2>&1
timeout 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o run -- python3 $R/scripts/profile_b16.py > $O/pmc2.log 2>&1
cd $R
python3 scripts/summarize_profiles.py pmc $O/b16_pmc_summary.csv $(find $O/pmc_* -name "*counter_collection.csv")
rm -rf $O/pmc_*
grep -v "at::native" $O/b16_pmc_summary.csv | grep -v "pack_"
