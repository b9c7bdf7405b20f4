export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -12
bash tools/ab.sh r03o_ab2 2 20 2 base.so chunk.so 2>&1 | tail -5
