"""CPU oracle for the DFCNN(+SE)+CTC / Transformer hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``asr_dfcnn_transformer_amd/`` may
import this package; only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` do, and only as the checker.

PARITY UNPINNED: the reference (786440445/ASR_DFCNN_Transformer) ships no
tests, golden vectors or fixtures for this path and its
arithmetic lives in third-party packages that are not installed here
(tensorflow 1.x, Keras 2.3.1, python_speech_features 0.6, scikit-learn).
Each function restates the published algorithm of the dependency named in
its docstring and cites the reference call site (file:line under
/root/reference).  The restatement is float64 numpy; every piece is
cross-checked in tests/ against an independent implementation available
offline (torch-CPU autograd, scipy.fft, sklearn.preprocessing.scale).

Pinned to the reference itself since round 5 (the pieces of it that run here
without TensorFlow, outputs committed under tests/golden/ with the scripts
that made them): fbank.build_LFR_features and ctc.get_edit_distance_difflib
against the reference's own util/utils.py functions
(tests/golden/reference_utils.npz, tests/test_reference_utils_golden.py).
Everything that is arithmetic of the network, fbank, CTC or Adam stays
unpinned as said above.
"""
