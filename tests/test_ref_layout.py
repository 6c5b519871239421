"""The byte layout gpsbb_fill_block_ref is tested with (conftest.REF_CHANNEL_DTYPE) against the REAL channel_t:
a C program including /root/reference/plutogpssim.h prints offsetof() of every field the entry point touches, plus
sizeof(channel_t).  Build container only (the header does not travel)."""
import os
import subprocess

import pytest

from conftest import REF_CHANNEL_DTYPE, REF_CHANNEL_FIXED_DTYPE

HDR = "/root/reference/plutogpssim.h"
FIELDS = ["prn", "ca", "f_carr", "f_code", "carr_phase", "code_phase", "g0", "sbf", "dwrd", "iword", "ibit", "icode",
          "dataBit", "codeCA", "azel", "rho0"]


@pytest.mark.skipif(not os.path.exists(HDR), reason="no /root/reference here")
@pytest.mark.parametrize("fixed", [False, True])
def test_channel_t_offsets(tmp_path, fixed):
    """fixed: the header with line 12 (`#define FLOAT_CARR_PHASE`) taken out in a temporary copy — the reference's other
    channel_t (h:160-161), the layout gpsbb_fill_block_ref_fixed reads."""
    global HDR, FIELDS
    hdr, fields, dtype = HDR, list(FIELDS), REF_CHANNEL_DTYPE
    if fixed:
        lines = open(HDR).read().splitlines(True)
        assert lines[11].startswith("#define FLOAT_CARR_PHASE")
        hdr = str(tmp_path / "plutogpssim_fixed.h")
        open(hdr, "w").write("".join(lines[:11] + lines[12:]))
        fields.insert(fields.index("carr_phase") + 1, "carr_phasestep")
        dtype = REF_CHANNEL_FIXED_DTYPE
    src = tmp_path / "off.c"
    body = "\n".join('    printf("%s %%zu\\n", offsetof(channel_t, %s));' % (f, f) for f in fields)
    src.write_text('#include <stdbool.h>\n#include <limits.h>\n#include <stddef.h>\n#include <stdint.h>\n#include <stdio.h>\n'
                   '#include <sys/types.h>\n#include "%s"\nint main(void) {\n%s\n'
                   '    printf("sizeof %%zu\\n", sizeof(channel_t));\n'
                   '    printf("sizeof_dwrd_elem %%zu\\n", sizeof(((channel_t *)0)->dwrd[0]));\n'
                   '    printf("sizeof_carr_phase %%zu\\n", sizeof(((channel_t *)0)->carr_phase));\n    return 0;\n}\n' % (hdr, body))
    exe = tmp_path / "off"
    subprocess.check_call(["gcc", "-std=c11", "-D_GNU_SOURCE", str(src), "-o", str(exe)])
    got = dict(l.split() for l in subprocess.check_output([str(exe)], text=True).splitlines())
    name = {"g0": "g0_week"}
    for f in fields:
        assert int(got[f]) == dtype.fields[name.get(f, f)][1], f
    assert int(got["sizeof"]) == dtype.itemsize
    assert int(got["sizeof_dwrd_elem"]) == 8
    assert int(got["sizeof_carr_phase"]) == (4 if fixed else 8)
