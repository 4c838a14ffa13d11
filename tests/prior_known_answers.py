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


# ---- collision avoidance: /root/reference/test/domains/priors/CollisionAvoidancePriorTests.cpp --------------------------------
CA_DOWN, CA_STAY, CA_UP = 0, 1, 2      # CollisionAvoidance.hpp: MOVE_DOWN, STAY, MOVE_UP
BLOCK_MOVE_PROB = .5                   # CollisionAvoidance.hpp (the probability that an obstacle stays where it is)


def check_ca_flat(counts, W, H):
    """the table prior with one obstacle and no noise (:15-131): transitionExpectation(state, STAY) -- expectedMult of the row, float
    sum and float division -- is EXACTLY .5 for the obstacle staying and .25 for each neighbour row, .75 / .25 at the two edges,
    whatever counts_total (the reference draws it from 3..19); state index = (x H + y) H + obstacle row"""
    S, A = W * H * H, 3
    phi = np.asarray(counts, np.float32)[:S * A * S].reshape(S, A, S)
    st = lambda x, y, b: (x * H + y) * H + b

    def expectation(s):
        row = phi[s, CA_STAY]
        total = np.float32(0)
        for v in row:                       # expectedMult (random.cpp:257-279): a float running sum
            total = np.float32(total + v)
        return row / total
    for x in range(1, W):
        for y in range(1, H):
            for b in range(1, H - 1):                                                                    # :52-79
                e = expectation(st(x, y, b))
                assert e[st(x - 1, y, b)] == BLOCK_MOVE_PROB
                assert e[st(x - 1, y, b + 1)] == (1 - BLOCK_MOVE_PROB) / 2 and e[st(x - 1, y, b - 1)] == (1 - BLOCK_MOVE_PROB) / 2
            e = expectation(st(x, y, 0))                                                                 # :81-102
            assert e[st(x - 1, y, 1)] == (1 - BLOCK_MOVE_PROB) / 2 and e[st(x - 1, y, 0)] == (1 + BLOCK_MOVE_PROB) / 2
            e = expectation(st(x, y, H - 1))                                                             # :104-127
            assert e[st(x - 1, y, H - 2)] == (1 - BLOCK_MOVE_PROB) / 2 and e[st(x - 1, y, H - 1)] == (1 + BLOCK_MOVE_PROB) / 2


def ca_factored_rows(blob, W, H, n):
    """row(a, f, v): the Dirichlet row of transition node (a, f) at value v of its one parent, normalised -- the blob of the prior with
    the correct graph (no structure prior: every transition node has itself as its only parent, its rows are all it stores)"""
    FS = 2 + n
    sizes = [W, H] + [H] * n
    b = np.asarray(blob, np.float32)

    def row(a, f, v):
        off = 0
        for aa in range(3):
            for ff in range(FS):
                if (aa, ff) == (a, f):
                    r = b[off + v * sizes[f]: off + (v + 1) * sizes[f]]
                    return r / r.sum()
                off += sizes[ff] * sizes[ff]
    return row


def check_ca_factored(blob, W, H, n, approx):
    """the factored prior, no noise (:215-340): obstacle rows expect {.25, .5, .25} in the middle and {.75, .25} at the edges, the agent
    always moves one column, its row follows the action and stops at the walls"""
    row = ca_factored_rows(blob, W, H, n)
    for a in range(3):
        for f in range(2, 2 + n):
            for pos in range(1, H - 1):                                                                  # :243-262
                e = row(a, f, pos)
                assert e[pos] == approx(BLOCK_MOVE_PROB) and e[pos + 1] == approx(BLOCK_MOVE_PROB * .5) and e[pos - 1] == approx(BLOCK_MOVE_PROB * .5)
            assert row(a, f, 0)[0] == approx(.5 * (1 + BLOCK_MOVE_PROB)) and row(a, f, 0)[1] == approx(.5 * (1 - BLOCK_MOVE_PROB))          # :264-279
            assert row(a, f, H - 1)[H - 1] == approx(.5 * (1 + BLOCK_MOVE_PROB)) and row(a, f, H - 1)[H - 2] == approx(.5 * (1 - BLOCK_MOVE_PROB))
        for x in range(1, W):
            assert row(a, 0, x)[x - 1] == 1.0                                                            # :296-301
    for y in range(1, H - 1):                                                                            # :307-323
        assert row(CA_UP, 1, y)[y + 1] == 1.0 and row(CA_STAY, 1, y)[y] == 1.0 and row(CA_DOWN, 1, y)[y - 1] == 1.0
    assert row(CA_UP, 1, H - 1)[H - 1] == 1.0 and row(CA_DOWN, 1, 0)[0] == 1.0                           # :325-329
