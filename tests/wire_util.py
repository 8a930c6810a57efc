"""Rebuilds the calldata of a tests/golden/wire_cases.json case: canonical encoding of the named method's arguments (by
the encoder passed in: oracle, host library or spec model) followed by the case's byte edits."""
H = bytes.fromhex


def apply_ops(calldata, ops):
    b = bytearray(calldata)
    for op in ops:
        if op[0] == 'patch':
            v = H(op[2]); b[op[1]:op[1] + len(v)] = v
        elif op[0] == 'truncate':
            del b[op[1]:]
        elif op[0] == 'append':
            b += H(op[1])
        elif op[0] == 'insert':
            b[op[1]:op[1]] = H(op[2])
    return bytes(b)


def calldata_of(case, enc_verify, enc_integrity, enc_sp1):
    m = case['method']
    if m == 'verify':
        cd = enc_verify(H(case['seal']), H(case['a']), H(case['b']))
    elif m == 'verify_integrity':
        cd = enc_integrity(H(case['seal']), H(case['a']))
    elif m == 'verify_proof':
        cd = enc_sp1(H(case['a']), H(case['pv']), H(case['seal']))
    else:
        cd = H(case['raw'])
    cd = apply_ops(cd, case['ops'])
    assert len(cd) == case['calldata_len'], case['name']
    return cd
