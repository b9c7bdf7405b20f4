export TMPDIR=/tmp
mkdir -p gpurun_out/r06k
timeout 1200 python bench.py --config 2 --soak-seconds 300 --no-serving --no-end-to-end --cpu-budget 5 --full-json gpurun_out/r06k/cfg2_soak_full.json > gpurun_out/r06k/cfg2_soak_line.json 2>/dev/null; echo rc=$?
python -c "
import json; d=json.load(open('gpurun_out/r06k/cfg2_soak_full.json')); print(d['ms_per_step'], d['value'], d['sustained'], d['cpu_baseline']['gpu_verdicts_identical_to_cpu'])"
rocm-smi --showclocks --showtemp --showpower 2>/dev/null | grep -v "^=\|^$" | head -20 > gpurun_out/r06k/rocm_smi_after_soak.txt; cat gpurun_out/r06k/rocm_smi_after_soak.txt | head -12
