"""bench.py against another build of the library (development A/B on one box): LIB=<path to libasrhip.so> python tools/bench_with_lib.py [bench args]"""
import os, sys, runpy
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from asr_dfcnn_transformer_amd import _lib
if os.environ.get('LIB'):
    _lib.LIB_PATH = os.path.abspath(os.environ['LIB'])
sys.argv = [os.path.join(ROOT, 'bench.py')] + sys.argv[1:]
runpy.run_path(os.path.join(ROOT, 'bench.py'), run_name='__main__')
