"""Trapdoor verification keys for parity tests: keys whose discrete logs are known, so VALID proofs exist for arbitrary public
inputs (the reference leaves proof generation as a TODO, examples/risc0-verifier/examples/interact.rs:110)."""
import spec_model as m


def generic_cases(rng):
    """(name, vm, vk dict, (a, b, c), signals, expected) for trapdoor keys: valid proofs with arbitrary public inputs."""
    out = []
    for vm, n_ic in (('risc0', 6), ('sp1', 3), ('sp1', 1), ('risc0', 2)):
        vk, td = m.trapdoor_vk(rng, n_ic)
        for j in range(2):
            sig = [rng.randrange(m.R) for _ in range(n_ic - 1)]
            if j == 1 and sig:
                sig[0] = rng.choice([0, 1, m.R - 1])
            prf = m.trapdoor_prove(rng, td, sig, vm)
            out.append(('%s n_ic=%d valid %d' % (vm, n_ic, j), vm, vk, prf, sig, True))
        if n_ic > 1:
            bad = list(sig); bad[-1] = (bad[-1] + 1) % m.R
            out.append(('%s n_ic=%d wrong signal' % (vm, n_ic), vm, vk, prf, bad, False))
            over = list(sig); over[0] = m.R
            out.append(('%s n_ic=%d signal = R' % (vm, n_ic), vm, vk, prf, over, False))
        out.append(('%s n_ic=%d other vm convention' % (vm, n_ic), 'sp1' if vm == 'risc0' else 'risc0', vk, prf, sig, False))
    # degenerate keys
    vk, td = m.trapdoor_vk(rng, 3)
    sig = [rng.randrange(m.R) for _ in range(2)]
    prf = m.trapdoor_prove(rng, td, sig, 'sp1')
    broken = dict(vk, ic=[vk['ic'][0], (vk['ic'][1][0], vk['ic'][1][1] ^ 1), vk['ic'][2]])        # IC[1] off curve
    out.append(('IC point off curve: every proof fails', 'sp1', broken, prf, sig, False))
    vk2 = dict(vk, ic=[vk['ic'][0], (0, 0), vk['ic'][2]])                                              # IC[1] = infinity
    td2 = dict(td, ic=[td['ic'][0], 0, td['ic'][2]])
    prf2 = m.trapdoor_prove(rng, td2, sig, 'sp1')
    out.append(('IC[1] = infinity', 'sp1', vk2, prf2, sig, True))
    vk3 = dict(vk, gamma2=((0, 0), (0, 0)))                                                            # gamma = infinity
    td3 = dict(td, gamma=0)
    prf3 = m.trapdoor_prove(rng, td3, sig, 'sp1')
    out.append(('gamma2 = infinity', 'sp1', vk3, prf3, sig, True))
    return out


