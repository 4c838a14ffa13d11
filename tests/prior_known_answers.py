"""The known answers of the reference's own prior tests for the factored tiger problem
(/root/reference/test/domains/priors/TigerPriorTest.cpp:230-376 flat BAPOMDPPrior, :378-591 FBAPOMDPPrior with and without
`uniform` structure noise at three noise levels, :593-617 fully connected, :619-655 match-uniform), as checks on a count
blob in the layout the oracle and the engine share (include/fba_hip.h: BAFlatModel order phi[s][a][s'], psi[a][s'][o]; factored:
T(a, f) nodes a-major, then O(a, 0), each node with room for its largest parent set, then the parent-set word of the
listen observation node).  Used against the oracle (tests/test_oracle_golden.py, CPU) and against the HIP engine through
fba_get_prior / fba_belief_get (tests/test_gpu_parity.py)."""
import numpy as np

KNOWN = np.float32(5000)        # TigerPriorTest.cpp:232, :380: `float known_counts = 5000`
TOTAL = 40.0                    # conf.counts_total = 40 (:255, :409)
NOISES = (0.0, 0.1, 0.2)        # :244, :396
OPEN_LEFT, OPEN_RIGHT, OBSERVE = 0, 1, 2   # FactoredTiger.hpp:38
LEFT, RIGHT = 0, 1                          # FactoredTiger.hpp:39


def acc(n):       # conf.counts_total * (.85f - n), float arithmetic as the REQUIREs evaluate it (:283-288)
    return np.float32(TOTAL) * (np.float32(.85) - np.float32(n))


def inacc(n):
    return np.float32(TOTAL) * (np.float32(.15) + np.float32(n))


def check_flat(counts, size, noise):
    """factored tiger BAPOMDPPrior (:230-376): S = 2 << size states, tiger LEFT iff s < S / 2 (FactoredTiger.cpp:26-29)"""
    S, A, O = 2 << size, 3, 2
    c = np.asarray(counts, np.float32)
    assert c.size == S * A * S + A * S * O
    phi, psi = c[:S * A * S].reshape(S, A, S), c[S * A * S:].reshape(A, S, O)
    for s in range(S):
        loc = LEFT if s < S // 2 else RIGHT
        assert psi[OBSERVE, s, loc] == acc(noise) and psi[OBSERVE, s, 1 - loc] == inacc(noise)          # :268-291
        assert np.all(psi[OPEN_LEFT, s] == KNOWN) and np.all(psi[OPEN_RIGHT, s] == KNOWN)                # :296-317
        expect = np.zeros(S, np.float32)
        expect[s] = KNOWN
        assert np.array_equal(phi[s, OBSERVE], expect)                                                   # :320-343
        assert np.all(phi[s, OPEN_LEFT] == KNOWN) and np.all(phi[s, OPEN_RIGHT] == KNOWN)                # :346-369


def factored_parts(blob, size):
    """(T open [2][FS][2], T listen [FS][2][2], O open [2][2], O listen room [2 << FS], parent-set word of the listen O node)"""
    FS = size + 1
    b = np.asarray(blob, np.float32)
    n_counts = 8 * FS + 4 + (2 << FS)
    assert b.size == n_counts + 1
    t_open = b[:4 * FS].reshape(2, FS, 2)
    t_listen = b[4 * FS:8 * FS].reshape(FS, 2, 2)
    o_open = b[8 * FS:8 * FS + 4].reshape(2, 2)
    o_listen = b[8 * FS + 4:n_counts]
    mask = int(b[n_counts:].view(np.uint32)[0])
    return t_open, t_listen, o_open, o_listen, mask


def check_factored(blob, size, noise, structure_noise):
    """factored tiger FBAPOMDPPrior (:378-591) on one state sampled from the prior"""
    t_open, t_listen, o_open, o_listen, mask = factored_parts(blob, size)
    FS = size + 1
    assert np.all(o_open == KNOWN)                            # :427-465: no parents, two parameters, uniform
    assert np.all(t_open == KNOWN)                            # :508-535: independent of any parent, uniform
    for f in range(FS):                                       # :537-580: listening keeps every feature where it is
        assert np.array_equal(t_listen[f], np.array([[KNOWN, 0], [0, KNOWN]], np.float32))
    if not structure_noise:                                   # :467-505 (skipped by the reference when structures may differ)
        assert mask == 1                                      # one parent: the tiger location (feature 0)
        assert np.array_equal(o_listen[:4], np.array([acc(noise), inacc(noise), inacc(noise), acc(noise)], np.float32))
        assert not o_listen[4:].any()                         # numParams() == 4: nothing beyond the four cells in use
    return mask
