#!/bin/bash
# the other BASELINE configs and shapes on one box (tools/sweep.py lines: step ms, assigns/s, dominant launch, chain kernel)
cd $GRAFT_REPO_ROOT
python tools/sweep.py CONFIG C2-driver-args --steps 20 --warmup 5
python tools/sweep.py CONFIG C3-shard-8192-as-4-calls --batch 2048 --chunks 4 --steps 20 --warmup 5
python tools/sweep.py CONFIG C3-shard-8192-as-2-calls --batch 4096 --chunks 2 --steps 20 --warmup 5
python tools/sweep.py CONFIG C3-shard-8192-one-call --batch 8192 --steps 20 --warmup 5
python tools/sweep.py CONFIG C3-shard-8192-serial --batch 8192 --steps 6 --warmup 2 --no-pipeline
python tools/sweep.py CONFIG C4-rsa4096-w32-4096 --workload rsa4096_w32_e65537 --batch 4096 --steps 4 --warmup 1
python tools/sweep.py CONFIG C4-rsa4096-w32-4096-serial --workload rsa4096_w32_e65537 --batch 4096 --steps 4 --warmup 1 --no-pipeline
python tools/sweep.py CONFIG C5-e2048bit-256 --workload rsa2048_e2048bit --batch 256 --steps 8 --warmup 2
python tools/sweep.py CONFIG C5-e2048bit-256-serial --workload rsa2048_e2048bit --batch 256 --steps 4 --warmup 1 --no-pipeline
python tools/sweep.py CONFIG rsa1024 --workload rsa1024_e65537 --steps 40 --warmup 4
python tools/sweep.py CONFIG rsa1536 --workload rsa1536_e65537 --steps 40 --warmup 4
python tools/sweep.py CONFIG rsa3072 --workload rsa3072_e65537 --steps 20 --warmup 3
python tools/sweep.py CONFIG rsa4096-w64 --workload rsa4096_e65537 --steps 20 --warmup 3
python tools/sweep.py CONFIG rsa2048-shared-modulus --steps 40 --warmup 4 --shared-modulus
python tools/sweep.py CONFIG rsa2048-verify --steps 40 --warmup 4 --verify
python tools/sweep.py CONFIG C1-one-signature-per-call --batch 1 --steps 200 --warmup 20 --no-pipeline
python tools/sweep.py CONFIG batch-64-per-call --batch 64 --steps 200 --warmup 20
