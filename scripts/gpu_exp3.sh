python scripts/diag_w70.py 2>&1 | tail -5
