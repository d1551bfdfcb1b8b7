#!/usr/bin/env python3
"""tests/golden/vote_table.npz: seeded sequences of votes and what the REFERENCE's own state machine
(improved_index_table_add, src/qv.cc:132-178, driven by oracle/ref_vote_replay.cc -- `make -C oracle vote_table`, build
container only) holds after each: best entry or none, its index, its uint8_t frequency, the ambiguity flag.
Sequences: few distinct positions so that ties and ambiguity flips are the rule; positions that share a slot of the 1009-slot
table; neighbour votes for positions no exact hit has opened (refused, qv.cc:134-139); the same k-mer position voting twice
(a position needs two DIFFERENT k-mer positions before it can lead, qv.cc:163-165); runs past 255 votes (the frequency is a
uint8_t).  The fixture is data: the sequences and the reference's answers."""
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
OUT = os.path.dirname(os.path.abspath(__file__))


def sequences(seed=31):
    rng = np.random.default_rng(seed)
    seqs = []
    for s in range(2400):
        kind = s % 6
        n = int(rng.integers(1, 30)) if kind < 4 else (int(rng.integers(258, 420)) if kind == 4 else int(rng.integers(30, 120)))
        nidx = int(rng.integers(1, 5)) if kind != 5 else int(rng.integers(3, 9))
        base = int(rng.integers(0, 2 ** 32 - 10 ** 6))
        if kind == 3 or kind == 5:
            idx_pool = base + 1009 * rng.choice(200, size=nidx, replace=False)           # one slot of the table
        else:
            idx_pool = base + rng.choice(5000, size=nidx, replace=False)
        idx_pool = (idx_pool % 2 ** 32).astype(np.uint32)
        index = idx_pool[rng.integers(0, nidx, size=n)]
        if kind == 4:
            index[:] = idx_pool[0] if rng.random() < 0.5 else index                       # one position past 255 votes
        kpos = (index.astype(np.int64) + 32 * rng.integers(0, 4 if kind != 4 else 40, size=n)).astype(np.uint32)
        neigh = (rng.random(n) < (0.5 if kind != 4 else 0.8)).astype(np.uint32)
        seqs.append((index, kpos, neigh))
    return seqs


if __name__ == "__main__":
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "vote_table"], stdout=subprocess.DEVNULL)
    seqs = sequences()
    work = "/tmp/vg_vote"
    os.makedirs(work, exist_ok=True)
    with open(os.path.join(work, "in.bin"), "wb") as f:
        f.write(np.uint32(len(seqs)).tobytes())
        for index, kpos, neigh in seqs:
            f.write(np.uint32(len(index)).tobytes())
            f.write(np.stack([index, kpos, neigh], axis=1).astype("<u4").tobytes())
    subprocess.check_call([os.path.join(ROOT, "oracle", "_ref", "ref_vote_replay"), os.path.join(work, "in.bin"), os.path.join(work, "out.bin")])
    res = np.fromfile(os.path.join(work, "out.bin"), dtype="<u4").reshape(-1, 4)
    assert len(res) == len(seqs)
    lens = np.array([len(s[0]) for s in seqs], dtype=np.uint32)
    np.savez_compressed(os.path.join(OUT, "vote_table.npz"), lens=lens, index=np.concatenate([s[0] for s in seqs]), kpos=np.concatenate([s[1] for s in seqs]),
                        neigh=np.concatenate([s[2] for s in seqs]).astype(np.uint8), result=res)
    print("%d sequences, %d votes; with a best entry: %d, ambiguous: %d, best freq < 20 after > 255 votes for it: %d" % (
        len(seqs), lens.sum(), int(res[:, 0].sum()), int(res[:, 3].sum()), int(((res[:, 2] < 20) & (lens > 300) & (res[:, 0] == 1)).sum())))
