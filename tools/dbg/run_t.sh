cd /root/repo; timeout 600 python -m pytest tests/test_round4.py -q -m gpu -x -k "lambda_gain_map" 2>&1 | tail -40
