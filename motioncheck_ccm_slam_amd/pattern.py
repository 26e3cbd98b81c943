"""The rBRIEF sampling pattern as a numpy array, parsed from include/ccm_orb_pattern.h."""
import os
import re

import numpy as np

_H = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include", "ccm_orb_pattern.h")
_txt = open(_H).read()
PATTERN = np.array([int(v) for v in re.findall(r"-?\d+", _txt.split("{")[1].split("}")[0])], np.int8)
SHA256 = re.search(r'CCM_ORB_PATTERN_SHA256 "([0-9a-f]+)"', _txt).group(1)
assert PATTERN.size == 1024
