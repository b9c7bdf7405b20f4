export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 600 python -m pytest tests -m gpu -x -q -k "collective_verify_matches or micro_batcher or fenced_shapes or exotic or cfg2_full_size" 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -3
