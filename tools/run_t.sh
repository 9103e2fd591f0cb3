T="tests/test_gpu_parity.py::test_model_vs_oracle_baseline_widths[lipo_c4-True]"
echo "== default"; python -m pytest "$T" -q -x 2>&1 | grep -E "passed|failed|AssertionError: \(" | head -3
echo "== KMEMSET"; EAGCN_G3_KMEMSET=1 python -m pytest "$T" -q -x 2>&1 | grep -E "passed|failed|AssertionError: \(" | head -3
echo "== NO_GEMM3"; EAGCN_NO_GEMM3=1 python -m pytest "$T" -q -x 2>&1 | grep -E "passed|failed|AssertionError: \(" | head -3
echo "== default again"; python -m pytest "$T" -q -x 2>&1 | grep -E "passed|failed|AssertionError: \(" | head -3
