#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/c12; O=gpurun_out/c12
timeout 600 python scripts/count_packs.py 2>&1 | grep -v "amdgpu.ids" | head -40 > $O/count_packs.txt
timeout 600 python scripts/aten_residue.py 2>&1 | grep -v "amdgpu.ids" > $O/aten_residue.txt
export TMPDIR=/tmp; R=$PWD; cd /tmp
timeout 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $R/$O/pmc_sq -o run -- python3 $R/scripts/profile_w4.py > $R/$O/pmc3.log 2>&1
timeout 200 rocprofv3 --pmc GRBM_GUI_ACTIVE --output-format csv -d $R/$O/pmc_grbm -o run -- python3 $R/scripts/profile_w4.py > $R/$O/pmc4.log 2>&1
cd $R
python3 scripts/summarize_profiles.py pmc $O/k1_pmc_summary.csv $(find $O/pmc_* -name "*counter_collection.csv")
grep "wino4" $O/k1_pmc_summary.csv
cat $O/count_packs.txt | head -30; cat $O/aten_residue.txt | head -60
