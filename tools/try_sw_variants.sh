run() { timeout -k 5 150 python bench.py --config sw --no-cpu-baseline 2>&1 | tail -1 | grep -o '"value": [0-9.]*\|"kernel_ms": [0-9.]*' | tr '\n' ' '; echo; }
echo "=== baseline"; run
for v in rovaca_amd/lib/variants/libsw_*.so; do echo "=== $v"; ROVACA_AMD_SW_LIB=$PWD/$v run; done
