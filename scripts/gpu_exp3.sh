python scripts/diag_w130.py 2>&1 | tail -9
