#!/bin/bash
# soak: 150 GAN iterations of train.py on synthetic data in both precisions (captured as a hipGraph after two eager steps); losses must stay finite
mkdir -p gpurun_out/c45
for prec in fp32 bf16; do
  timeout 900 python train.py --phase train --synthetic 2400 --batch_size 16 --patch_size 48 --num_epochs 1 --max_iters 150 --learning_rate 5e-7 --check_point gpurun_out/c45/ck_$prec --precision $prec 2>&1 | grep -v amdgpu | tail -6 > gpurun_out/c45/train_$prec.txt
  echo "== $prec"; cat gpurun_out/c45/train_$prec.txt
done
rm -rf gpurun_out/c45/ck_*
