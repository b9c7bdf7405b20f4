#!/bin/bash
# the whole protocol file in its own order (the new ReadEntity test runs last, after the tests that fill the certificate caches)
timeout 85 python -m pytest -q -x --durations=5 -p no:cacheprovider tests/test_gpu_protocol.py > gpurun_out/r4af_protocol.txt 2>&1
echo "rc=$?" >> gpurun_out/r4af_protocol.txt
tail -15 gpurun_out/r4af_protocol.txt
