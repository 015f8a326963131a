python -m pytest tests -m gpu -x -q 2>&1 | tail -2
python tools/kbench.py sqiswap 65536 32 4 | cut -c1-260
bash tools/pmc_steady.sh sqiswap 2 8 2>&1 | grep -E "LDS_BANK|LDS_IDX|duration|VALU-active"
bash tools/pmc_steady.sh sqiswap 1 8 2>&1 | grep -E "LDS_BANK|LDS_IDX|duration|VALU-active"
bash tools/pmc_steady.sh sqiswap 3 8 2>&1 | grep -E "LDS_BANK|LDS_IDX|duration|VALU-active"
