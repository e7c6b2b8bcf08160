#!/usr/bin/env python
"""Generates bayespy_amd/csrc/vmp_exp2_table.h: 2^(j/256) for j = 0..255, each entry the double
nearest to the 60-digit decimal value (hex float literals, so the compiler cannot re-round)."""
import os
from decimal import Decimal, getcontext

getcontext().prec = 60
ln2 = Decimal(2).ln()
vals = [float((ln2 * Decimal(j) / Decimal(256)).exp()) for j in range(256)]
lines = ['    ' + ', '.join(float.hex(v) for v in vals[i:i + 4]) + ',' for i in range(0, 256, 4)]
out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'bayespy_amd',
                   'csrc', 'vmp_exp2_table.h')
with open(out, 'w') as f:
    f.write('// 2^(j/256), j = 0..255, correctly rounded (generated with 60-digit decimal '
            'arithmetic:\n// tools/gen_exp2_table.py).  Table of exp_tab() in vmp_gmm.hip.\n')
    f.write('static __device__ const double VMP_EXP2_TAB[256] = {\n' + '\n'.join(lines) + '\n};\n')
