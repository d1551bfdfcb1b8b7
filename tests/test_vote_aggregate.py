"""The device keeps a pass's exact contexts BY VOTE KEY and computes the vote from per-key totals (vargeno_amd/csrc/vg_wave.h,
stage C) instead of replaying improved_index_table_add (qv.cc:132-178) vote by vote.  That rests on a claim about the
reference's state machine: after any sequence of votes in which no key exceeds 255 votes,
    * `best` exists iff some key has been voted for with two different k-mer positions (a "live" key, qv.cc:163-165);
    * `best` is a live key of maximal frequency, where a key's frequency counts every vote cast from the moment an exact
      context opened it (neighbour votes for a key nobody has opened are refused, qv.cc:134-139);
    * `ambiguous` is set iff a second live key has that frequency.
So whether a pass is processed (qv.cc:1375: best && freq > 1 && !ambiguous) and at which position does not depend on the
order of the votes beyond "was the key open yet".  This file checks the claim against the reference's OWN state machine:
tests/golden/vote_table.npz holds 2 400 seeded vote sequences and what oracle/ref_vote_replay.cc (which #includes the
reference's src/qv.cc) held after each."""
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def aggregate_vote(index, kpos, neigh):
    """Per-key totals -> (has_best, best_index or None when tied, best_freq, ambiguous, max votes any key received)."""
    freq, seen = {}, {}
    for i, k, ng in zip(index.tolist(), kpos.tolist(), neigh.tolist()):
        if i not in freq:
            if ng:
                continue                                  # refused: no exact context has opened this key yet
            freq[i], seen[i] = 0, set()
        freq[i] += 1
        seen[i].add(k)
    live = [i for i in freq if len(seen[i]) >= 2]
    most = max(freq.values()) if freq else 0
    if not live:
        return False, None, 0, False, most
    top = max(freq[i] for i in live)
    winners = [i for i in live if freq[i] == top]
    return True, (winners[0] if len(winners) == 1 else None), top, len(winners) > 1, most


def test_vote_outcome_is_a_function_of_per_key_totals():
    z = np.load(os.path.join(GOLDEN, "vote_table.npz"))
    lens, index, kpos, neigh, want = (z[k] for k in ("lens", "index", "kpos", "neigh", "result"))
    ends = np.cumsum(lens.astype(np.int64))
    checked = decided = tied = 0
    for s in range(len(lens)):
        a, b = int(ends[s] - lens[s]), int(ends[s])
        has, idx, f, amb, most = aggregate_vote(index[a:b], kpos[a:b], neigh[a:b])
        if most > 255:
            continue                                      # the uint8_t frequency wrapped: order matters there (the device's lists cannot get that far)
        w = tuple(int(x) for x in want[s])
        assert has == bool(w[0]), (s, has, w)
        if has:
            assert amb == bool(w[3]), (s, amb, w)
            assert f == w[2], (s, f, w)
            if not amb:
                assert idx == w[1], (s, idx, w)
                decided += 1
            else:
                tied += 1
        checked += 1
    assert checked > 1900 and decided > 1000 and tied > 100, (checked, decided, tied)


def test_vote_outcome_does_not_depend_on_the_order_within_a_chunk():
    """The device's rule as it applies it: contexts arrive chunk by chunk (a chunk's exact contexts before its neighbour
    contexts); shuffling the exact contexts of a chunk among themselves, and its neighbour contexts among themselves, leaves
    the aggregate outcome unchanged (it only asks whether a key was open when a neighbour voted) -- checked against a direct
    replay of the state machine as the oracle restates it."""
    from oracle import oracle as O

    rng = np.random.default_rng(5)
    for trial in range(300):
        n_chunks = int(rng.integers(2, 6))
        pool = rng.choice(4000, size=int(rng.integers(1, 6)), replace=False).astype(np.uint32) + 100000
        seq = []
        for c in range(n_chunks):
            ex = [(int(k), int(k) + 32 * c, 0) for k in pool[rng.random(len(pool)) < 0.6]]
            ne = [(int(k), int(k) + 32 * c, 1) for k in pool[rng.random(len(pool)) < 0.3] for _ in range(int(rng.integers(1, 3)))]
            seq.append((ex, ne))
        outcomes = set()
        for rep in range(4):
            flat = []
            for ex, ne in seq:
                ex2, ne2 = list(ex), list(ne)
                if rep:
                    rng.shuffle(ex2)
                    rng.shuffle(ne2)
                flat += ex2 + ne2
            if not flat:
                continue
            a = np.array(flat, dtype=np.uint32)
            got = O.vote_replay(a[:, 0], a[:, 1], a[:, 2].astype(np.uint8))
            has, idx, f, amb, _ = aggregate_vote(a[:, 0], a[:, 1], a[:, 2])
            assert bool(got[0]) == has and (not has or (bool(got[3]) == amb and int(got[2]) == f and (amb or int(got[1]) == idx))), (trial, rep, got, (has, idx, f, amb))
            outcomes.add((has, idx, f, amb))
        assert len(outcomes) <= 1
