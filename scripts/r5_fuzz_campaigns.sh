#!/bin/bash
# builder-run randomised parity campaigns on the round's final kernels (beyond the fixed-seed slice in tests/test_gpu_fuzz.py)
O=gpurun_out/r5_fuzz2.log
: > $O
timeout -k 10 400 python scripts/fuzz_parity.py --cases 60 --seed 5311 --graphs --log $O
timeout -k 10 400 python scripts/fuzz_parity.py --cases 50 --seed 5312 --worlds 5,6,7,8 --sizes 3000,12000,40000 --graphs --log $O
timeout -k 10 300 python scripts/fuzz_parity.py --cases 50 --seed 5313 --worlds 1,2,4,8 --max-steps 10 --log $O
timeout -k 10 300 python scripts/fuzz_parity.py --cases 16 --seed 5314 --worlds 1,2,8 --sizes 90000,262144 --max-steps 5 --log $O
grep -c "^case" $O; grep "fuzz done" $O; grep -c MISMATCH $O
