import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from flamo_amd import _lib
var = int(sys.argv[1]); sys.argv = [sys.argv[0]] + sys.argv[2:]
_lib.lib().fl_debug_set_mimo_variant(var, 0)
import bench_fdn
bench_fdn.main()
