mkdir -p gpurun_out/r06z
python -m pytest tests -q -m gpu > gpurun_out/r06z/suite_full.txt 2>&1
grep -E "passed|failed|error" gpurun_out/r06z/suite_full.txt | tail -3
