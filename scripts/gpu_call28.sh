#!/bin/bash
export TMPDIR=/tmp; R=$PWD; O=$R/gpurun_out/c28; mkdir -p $O
timeout 300 python -m pytest tests/test_bf16_gpu.py -x -q -k "wgrad" 2>&1 | tail -3
timeout 300 python scripts/bf16_time.py 2>&1 | grep -E "^wgrad" | tee $O/time.txt
cd /tmp
timeout 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o run -- python3 $R/scripts/profile_b16.py > $O/pmc1.log 2>&1
timeout 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o run -- python3 $R/scripts/profile_b16.py > $O/pmc2.log 2>&1
cd $R
python3 scripts/summarize_profiles.py pmc $O/b16_pmc_fw.csv $(find $O/pmc_* -name "*counter_collection.csv")
rm -rf $O/pmc_*
grep -v "at::native" $O/b16_pmc_fw.csv | grep -v pack_
