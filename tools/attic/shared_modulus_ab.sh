python -m pytest tests -m gpu -x -q 2>&1 | tail -3
for rep in 1 2; do
python tools/sweep.py H2R_TAG distinct-serial --steps 40 --warmup 4 --no-pipeline
python tools/sweep.py H2R_TAG shared-serial --steps 40 --warmup 4 --no-pipeline --shared-modulus
python tools/sweep.py H2R_TAG distinct-pipe --steps 40 --warmup 4
python tools/sweep.py H2R_TAG shared-pipe --steps 40 --warmup 4 --shared-modulus
done
python tools/sweep.py H2R_TAG b8192-pipe --steps 10 --warmup 3 --batch 8192
python tools/sweep.py H2R_TAG b8192-serial --steps 10 --warmup 3 --batch 8192 --no-pipeline
