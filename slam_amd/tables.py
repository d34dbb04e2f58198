"""Python view of include/slamfe_tables.h (parsed from the header, so there is
one source of truth for the constants)."""
from __future__ import annotations

import os
import re

import numpy as np

_HDR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include",
                    "slamfe_tables.h")


def _macro_body(text: str, name: str) -> str:
    m = re.search(r"#define\s+" + name + r"\s+(.*?)(?<!\\)\n", text, re.S)
    if not m:
        raise KeyError(name)
    return m.group(1).replace("\\\n", " ")


def _ints(body: str):
    return [int(x) for x in re.findall(r"-?\d+", body)]


_text = open(_HDR).read()
PATTERN = np.array(_ints(_macro_body(_text, "SLAMFE_PATTERN_INIT")), dtype=np.int32).reshape(256, 4)
PATTERN_CRC32 = int(re.search(r"SLAMFE_PATTERN_CRC32\s+(0x[0-9a-fA-F]+)", _text).group(1), 16)
FAST_RING = np.array(_ints(_macro_body(_text, "SLAMFE_FAST_RING_INIT")), dtype=np.int32).reshape(16, 2)
UMAX = np.array(_ints(_macro_body(_text, "SLAMFE_UMAX_INIT")), dtype=np.int32)
GAUSS7 = np.array(_ints(_macro_body(_text, "SLAMFE_GAUSS7_INIT")), dtype=np.int32)
