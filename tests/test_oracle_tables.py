"""Checks the oracle's constant tables against the reference's source text.  Runs only where
/root/reference exists (this container); reading the reference as text is study, not execution."""
import os
import re

import numpy as np
import pytest

REF = "/root/reference/NVorbis"
pytestmark = pytest.mark.skipif(not os.path.isdir(REF), reason="reference tree not present")


def test_inverse_db_table_matches_reference_literals(oracle):
    src = open(os.path.join(REF, "Floor1.cs")).read()
    body = re.search(r"inverse_dB_table =\s*\{(.*?)\};", src, re.S).group(1)
    vals = np.array([np.float32(t.strip().rstrip("f")) for t in body.replace("\n", " ").split(",") if t.strip()],
                    dtype=np.float32)
    assert vals.shape == (256,)
    assert np.array_equal(vals.view(np.uint32), oracle.inverse_db_table().view(np.uint32))


def test_clip_constants_match_reference(oracle):
    src = open(os.path.join(REF, "Utils.cs")).read()
    assert "LowerClip = -0.99999994f" in src and "UpperClip = 0.99999994f" in src
