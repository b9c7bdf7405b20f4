#!/bin/bash
# last look at the final library: the two certificate tests, the plain-C caller, smoke()
T=tests/test_gpu_protocol.py
timeout 30 python -m pytest -q -x -p no:cacheprovider $T::test_entity_verification_and_quorum_certificate $T::test_read_entity_shape_by_shape_on_the_gpu \
  tests/test_c_harness.py::test_c_caller_verifies_on_the_gpu > gpurun_out/r4ag.txt 2>&1
echo "rc=$?" >> gpurun_out/r4ag.txt
timeout 20 python __graft_entry__.py smoke >> gpurun_out/r4ag.txt 2>&1
echo "smoke rc=$?" >> gpurun_out/r4ag.txt
tail -8 gpurun_out/r4ag.txt
