"""ORACLE (test infrastructure): IndexTTS.remove_long_silence, indextts/infer.py:446-497, restated on numpy.

Pinned by tests/golden/host_logic.json: outputs of the reference's own method (imported in the build container with
inert stand-ins for the packages it only touches at import time, tests/golden/make_host_golden.py), checked in
tests/test_host_logic.py::test_host_helpers_match_reference_fixture.
"""
import numpy as np


def remove_long_silence(codes: np.ndarray, stop_mel_token=8193, silent_token=52, max_consecutive=30):
    codes = np.asarray(codes)
    rows, lens, fixed = [], [], False
    for code in codes:
        stops = np.nonzero(code == stop_mel_token)[0]
        n = int(stops[0]) if stops.size else code.shape[0]          # :461-465
        if int((code == silent_token).sum()) > max_consecutive:     # :468-469 (counts over the WHOLE row)
            keep, run = [], 0
            for k in range(n):                                      # :472-478
                if code[k] != silent_token:
                    keep.append(k)
                    run = 0
                elif run < 10:
                    keep.append(k)
                    run += 1
            rows.append(code[keep])
            lens.append(len(keep))
            fixed = True
        else:
            rows.append(code[:n])
            lens.append(n)
    if fixed:                                                       # :487-491
        if len(rows) > 1:
            m = max(len(r) for r in rows)
            out = np.full((len(rows), m), stop_mel_token, dtype=codes.dtype)
            for i, r in enumerate(rows):
                out[i, : len(r)] = r
            codes = out
        else:
            codes = rows[0][None]
    max_len = max(lens)                                             # :493-495
    if max_len < codes.shape[1]:
        codes = codes[:, :max_len]
    return codes, np.asarray(lens, dtype=np.int64)
