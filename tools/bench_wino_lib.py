"""tools/bench_wino_ab.py against another build of the library: LIB=<path to libasrhip.so> python tools/bench_wino_lib.py"""
import os, sys, runpy
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from asr_dfcnn_transformer_amd import _lib
if os.environ.get('LIB'):
    _lib.LIB_PATH = os.path.abspath(os.environ['LIB'])
runpy.run_path(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'bench_wino_ab.py'), run_name='__main__')
