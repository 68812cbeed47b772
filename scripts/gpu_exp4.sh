timeout 900 python -m pytest tests -m gpu -q -x -k "decoder_tail or smooth or post_process" 2>&1 | tail -25
