#!/usr/bin/env python
"""One-off source clean-up (round 4): resolve compile-time ablation switches whose verdict is recorded (docs/measurements.md) to
their adopted values, at the preprocessor level.  Handles `#if X`, `#if !X`, `#ifdef X`, `#ifndef X`, `#else`, `#endif` blocks whose
condition is exactly one of the given macros, drops their `#ifndef X / #define X v / #endif` default blocks, and leaves every other
directive alone.  Remaining uses of a macro inside C++ expressions are replaced by its literal value.
    python tools/prune_switches.py <file> ...    (edits in place)"""
import re
import sys

FIXED = {  # macro: value (None = never defined)
    'GADAPT_ABL_FIXED_SCALE': None, 'GADAPT_ABL_S_NO_X': None, 'GADAPT_ABL_S_NO_GEMM': None, 'GADAPT_ABL_NO_GEMM': None,
    'GADAPT_ABL_NO_EDGEWS': None, 'GADAPT_ABL_NO_DA': None, 'GADAPT_ABL_NO_ACCUM': None, 'GADAPT_ABL_EDGEWS_LINEAR': None,
    'GADAPT_DA_BPREFETCH': 1, 'GADAPT_SPLIT_PK': 0, 'GADAPT_MFMA_INTERLEAVE': 1, 'GADAPT_T_MFMA_PRIO': 0,
    'GADAPT_STAGGER_T': 0, 'GADAPT_STAGGER_FWD': 0, 'GADAPT_T_TWO_BUFFERS': 0,
    # round 6: switches whose losing side has been measured at least twice (docs/measurements.md B-J)
    'GADAPT_DIAG_G_TILE0': None, 'GADAPT_DIAG_EWS_TARGET_ORDER': None,
    'GADAPT_T_DIFF': 1, 'GADAPT_T_STREAM': 0, 'GADAPT_S_STREAM_DXD': 0, 'GADAPT_S_ALTERNATE': 0, 'GADAPT_T_ALTERNATE': 1,
    'GADAPT_DA_IN_SOURCE': 0, 'GADAPT_DA_F16': 0, 'GADAPT_BWD_OUT4': 1, 'GADAPT_XC_COMPACT_KERNEL': 1, 'GADAPT_XC_ONE_KSTEP': 1,
    'GADAPT_PK_DOT': 1, 'GADAPT_GEMM_SPLIT': 1, 'GADAPT_PRESPLIT_F': 1, 'GADAPT_PRESPLIT_S': 1, 'GADAPT_PRESPLIT_T': 1,
    'GADAPT_SPLIT_F16_S': 1, 'GADAPT_SPLIT_F16_T': 0, 'GADAPT_SPLIT_F16_WIDE': 0, 'GADAPT_FPL': 8, 'GADAPT_FWD_ONE_WAVE': 0,
    'GADAPT_PRECISE_SOFTMAX': 1, 'GADAPT_S_STREAM': 2, 'GADAPT_DA_UNROLL': 2, 'GADAPT_S_RESIDENT_B': 1,
}


def cond(line):
    """(macro, truth) if the directive tests exactly one FIXED macro, else None."""
    m = re.match(r'\s*#\s*(ifdef|ifndef|if|elif)\s+(.*?)\s*(//.*|/\*.*)?$', line)
    if not m:
        return None
    kind, expr = m.group(1), m.group(2).strip()
    if kind in ('ifdef', 'ifndef'):
        if expr in FIXED:
            defined = FIXED[expr] is not None
            return expr, defined if kind == 'ifdef' else not defined
        return None
    if kind == 'elif':
        return ('ELIF', None) if any(k in expr for k in FIXED) else None
    neg = expr.startswith('!')
    name = expr[1:].strip() if neg else expr
    m2 = re.match(r'^defined\s*\(?\s*(\w+)\s*\)?$', name)
    if m2 and m2.group(1) in FIXED:
        val = FIXED[m2.group(1)] is not None
        return m2.group(1), (not val) if neg else val
    if name in FIXED and FIXED[name] is not None:
        val = bool(FIXED[name])
        return name, (not val) if neg else val
    return None


def prune(text):
    lines = text.split('\n')
    out, stack = [], []        # stack entries: None (foreign block) or [keep_now, taken]
    i = 0
    while i < len(lines):
        ln = lines[i]
        s = ln.strip()
        # default-definition block of a fixed macro: #ifndef X / #define X ... / #endif
        m = re.match(r'#\s*ifndef\s+(\w+)', s)
        if m and m.group(1) in FIXED and i + 2 < len(lines) and re.match(r'\s*#\s*define\s+' + m.group(1) + r'\b', lines[i + 1]):
            j = i + 1
            while not lines[j].strip().startswith('#endif'):
                j += 1
            if all(st is None or st[0] for st in stack):
                pass                                            # dropped
            i = j + 1
            continue
        if s.startswith('#if'):
            c = cond(ln)
            if c is None or c[0] == 'ELIF':
                stack.append(None)
                if all(st is None or st[0] for st in stack[:-1]):
                    out.append(ln)
            else:
                stack.append([c[1], c[1]])
            i += 1
            continue
        if s.startswith('#elif') and stack and stack[-1] is not None:
            raise SystemExit(f"#elif on a pruned block near line {i + 1}: handle by hand")
        if s.startswith('#else') and stack and stack[-1] is not None:
            st = stack[-1]
            st[0] = not st[1]
            i += 1
            continue
        if s.startswith('#endif') and stack:
            st = stack.pop()
            if st is None and all(x is None or x[0] for x in stack):
                out.append(ln)
            i += 1
            continue
        if all(st is None or st[0] for st in stack):
            out.append(ln)
        i += 1
    text = '\n'.join(out)
    for k, v in FIXED.items():
        if v is not None:
            text = re.sub(r'\b' + k + r'\b', str(v), text)
    return text


for path in sys.argv[1:]:
    src = open(path).read()
    new = prune(src)
    if new != src:
        open(path, 'w').write(new)
        print('pruned', path, len(src.split('\n')), '->', len(new.split('\n')))
