cd /root/repo
python tools/dbg/soak.py 2>&1 | tail -8
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
