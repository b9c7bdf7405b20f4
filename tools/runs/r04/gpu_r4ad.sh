#!/bin/bash
# the ReadEntity walk on the GPU (device classes 3 / 4, new validity rules) + every test that reads certificates; then cfg 3 with
# 19-bit DSA tables (27 table products per verification instead of 29)
T=tests/test_gpu_protocol.py
timeout 170 python -m pytest -x -q --durations=6 -p no:cacheprovider \
  $T::test_read_entity_shape_by_shape_on_the_gpu $T::test_entity_verification_and_quorum_certificate \
  $T::test_batcher_cert_verify_for_principals_outside_the_keyring $T::test_server_sign_verify $T::test_read_proof_and_register_sites \
  $T::test_dsa_certificates_from_requests_take_bounded_table_slots $T::test_http_wire_replay $T::test_audit_plain_storage_db \
  > gpurun_out/r4ad_tests.txt 2>&1
echo "rc=$?" >> gpurun_out/r4ad_tests.txt
tail -25 gpurun_out/r4ad_tests.txt
BFTKV_DSA_WBITS=19 timeout 75 python bench.py --config 3 --steps 5 --warmup 2 --no-cpu-baseline --soak-seconds 0 \
  > gpurun_out/r4ad_cfg3_19bit.json 2> gpurun_out/r4ad_cfg3_19bit.err
echo "bench rc=$?"
tail -c 600 gpurun_out/r4ad_cfg3_19bit.json
tail -3 gpurun_out/r4ad_cfg3_19bit.err
