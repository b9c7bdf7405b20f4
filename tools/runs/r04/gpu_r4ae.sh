#!/bin/bash
# second pass: the certificate tests with the by-issuer rule, then cfg 3 at the library's own table width (bench prices a DSA
# verification by bftkv_gpu_dsa_window_bits)
T=tests/test_gpu_protocol.py
timeout 100 python -m pytest -q --durations=6 -p no:cacheprovider \
  $T::test_read_entity_shape_by_shape_on_the_gpu $T::test_entity_verification_and_quorum_certificate \
  $T::test_batcher_cert_verify_for_principals_outside_the_keyring $T::test_server_sign_verify $T::test_read_proof_and_register_sites \
  $T::test_dsa_certificates_from_requests_take_bounded_table_slots $T::test_http_wire_replay $T::test_audit_plain_storage_db \
  > gpurun_out/r4ae_tests.txt 2>&1
echo "rc=$?" >> gpurun_out/r4ae_tests.txt
tail -30 gpurun_out/r4ae_tests.txt
timeout 60 python bench.py --config 3 --steps 5 --warmup 2 --no-cpu-baseline --soak-seconds 0 \
  > gpurun_out/r4ae_cfg3.json 2> gpurun_out/r4ae_cfg3.err
echo "bench rc=$?"
tail -c 900 gpurun_out/r4ae_cfg3.json
tail -3 gpurun_out/r4ae_cfg3.err
