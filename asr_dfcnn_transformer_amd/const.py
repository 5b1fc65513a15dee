"""Token ids and flags of the reference (util/const.py:35-41).  The machine-specific path
table of the reference (util/const.py:47-78) is out of scope: data roots are arguments."""
import os


class Const:
    IGNORE = -1
    PAD = 0
    SOS = 1
    EOS = 2
    PAD_FLAG = '<pad>'
    SOS_FLAG = '<sos>'
    EOS_FLAG = '</sos>'
    DictFolder = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'data')
