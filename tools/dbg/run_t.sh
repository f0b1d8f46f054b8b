cd /root/repo; python tools/dbg/probe_width.py 2>&1 | grep GB
