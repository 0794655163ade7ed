python -m pytest tests -m gpu -q -x -k "shim" > gpurun_out/pytest_gpu.log 2>&1; grep -E "passed|failed|Error|assert|^E " gpurun_out/pytest_gpu.log | head
