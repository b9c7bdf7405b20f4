export TMPDIR=/tmp
bash tools/profile_bench.sh r03i_cfg5 5 10 2>&1 | tail -1
bash tools/profile_bench.sh r03i_cfg3 3 6 2>&1 | tail -1
