"""A model of the lean slab step's protocol between ranks (include/sph.h, sph_slab_step; csrc: k_slab_head, k_rebuild_slab): threads
stand for ranks, Python lists for the peer-mapped blocks — two receive buffers per side (parity of the step), one arrival flag per
side that only grows (2 t for the update of step t, 2 t + 1 for the records of a rebuild step t), slot arrays for the MAX of the
rebuild word.  What the model checks is the ORDER argument of DESIGN.md 6, not the kernels: with arbitrary delays between a rank's
launches — a rank may sit between its head kernel and the rest of its step for longer than its neighbours need for a whole step —
  * every message a rank reads is the one of ITS step (never the neighbour's next one: the buffer of a parity is written again only
    two steps later, and nobody begins step t + 2 without everybody's word for step t + 1),
  * a wait for "at least" 2 t completes although the neighbour has already raised the flag to 2 (t + 1) — and the wait for EQUALITY that
    rounds 3-5 shipped does not (the defect tests/test_slab_c_host.py::test_c_host_lean_step_with_a_rank_held_up_between_its_launches
    reproduces on the GPU),
  * all ranks take the rebuild branch in the same steps.
Round 6: the same for the speculative lean step (run_model_speculative), in which the wait for the neighbours' update comes BEFORE
the word goes round.
No GPU, no library: the CPU suite."""
import random
import threading
import time

import pytest


class Block:      # what one rank exports
    def __init__(self, n):
        self.recv = {side: [None, None] for side in (0, 1)}      # [side][parity] = (kind, step, sender)
        self.flag = {0: 0, 1: 0}                                  # arrival flags: what came from the left / from the right
        self.slots = [[0] * n, [0] * n]                           # [parity][sender] = step << 2 | word


def run_model(n_ranks, steps, wait_at_least, seed, stall_rank=1, timeout=20.0, speculative=False):
    if speculative:
        return run_model_speculative(n_ranks, steps, seed, stall_rank, timeout)
    blocks = [Block(n_ranks) for _ in range(n_ranks)]
    errors, rebuilt = [], [[] for _ in range(n_ranks)]
    deadline = time.time() + timeout
    words = [[1 if random.Random(1000 * seed + t).random() < 0.2 else 0 for t in range(steps + 2)] for _ in range(n_ranks)]
    for r in range(1, n_ranks):      # (a rank's own criterion: independent draws)
        words[r] = [1 if random.Random(7919 * r + 31 * seed + t).random() < 0.1 else 0 for t in range(steps + 2)]

    def wait(pred, what):
        while not pred():
            if time.time() > deadline:
                raise TimeoutError(what)
            time.sleep(0)

    def rank(me):
        rng = random.Random(seed * 131 + me)
        left, right = (me - 1 if me > 0 else None), (me + 1 if me < n_ranks - 1 else None)
        try:
            for t in range(1, steps + 1):
                par = t & 1
                # head kernel: push this step's update, raise the neighbours' flags, exchange the word
                for nb, side_there in ((left, 1), (right, 0)):      # my message arrives at the neighbour's OTHER side
                    if nb is not None:
                        blocks[nb].recv[side_there][par] = ("update", t, me)
                        blocks[nb].flag[side_there] = 2 * t
                for q in range(n_ranks):
                    blocks[q].slots[par][me] = (t << 2) | words[me][t]
                for q in range(n_ranks):
                    wait(lambda q=q: blocks[me].slots[par][q] >> 2 == t, "rank %d step %d: word of rank %d" % (me, t, q))
                word = max(blocks[me].slots[par][q] & 3 for q in range(n_ranks))
                # ... and here a rank may be held up (time-slicing; a late launch)
                if me == stall_rank and t % 3 == 0:
                    time.sleep(0.002)
                elif rng.random() < 0.3:
                    time.sleep(rng.random() * 0.0003)
                ok = (lambda f, tag: f >= tag) if wait_at_least else (lambda f, tag: f == tag)
                if word == 0:      # the update-or-rebuild launch, update path: wait for my flags, unpack
                    for nb, side in ((left, 0), (right, 1)):
                        if nb is not None:
                            wait(lambda side=side: ok(blocks[me].flag[side], 2 * t), "rank %d step %d: update flag, side %d" % (me, t, side))
                            got = blocks[me].recv[side][par]
                            if got != ("update", t, nb):
                                errors.append((me, t, side, got))
                else:              # rebuild path: records to the neighbours, flags 2 t + 1, wait for mine, ingest
                    rebuilt[me].append(t)
                    for nb, side_there in ((left, 1), (right, 0)):
                        if nb is not None:
                            blocks[nb].recv[side_there][par] = ("records", t, me)
                            blocks[nb].flag[side_there] = 2 * t + 1
                    for nb, side in ((left, 0), (right, 1)):
                        if nb is not None:
                            wait(lambda side=side: ok(blocks[me].flag[side], 2 * t + 1), "rank %d step %d: records flag, side %d" % (me, t, side))
                            got = blocks[me].recv[side][par]
                            if got != ("records", t, nb):
                                errors.append((me, t, side, got))
                # density, force (the force pass leaves the next update in the send buffers: pushed by the next head)
                if rng.random() < 0.3:
                    time.sleep(rng.random() * 0.0002)
        except TimeoutError as e:
            errors.append(("timeout", str(e)))

    threads = [threading.Thread(target=rank, args=(r,)) for r in range(n_ranks)]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    return errors, rebuilt


def run_model_speculative(n_ranks, steps, seed, stall_rank=1, timeout=20.0):
    """The speculative lean step (round 6; sph_slab_set_speculative, both forms — four launches or the head fused into the density
    launch: the same order between the ranks): per step  push my update + wait for the neighbours' + unpack  |  density (whose jobs
    produce this rank's word)  |  gate: exchange the word, then nothing — or push my records, wait for theirs, ingest  |  force.
    The wait for the neighbours' update now comes BEFORE the word goes round, the records still after: the update of step t and the
    records of step t share the buffer of parity t, and nobody may write records over an update its neighbour has not unpacked yet."""
    blocks = [Block(n_ranks) for _ in range(n_ranks)]
    errors, rebuilt = [], [[] for _ in range(n_ranks)]
    deadline = time.time() + timeout
    words = [[1 if random.Random(7919 * r + 31 * seed + t).random() < 0.15 else 0 for t in range(steps + 2)] for r in range(n_ranks)]

    def wait(pred, what):
        while not pred():
            if time.time() > deadline:
                raise TimeoutError(what)
            time.sleep(0)

    def rank(me):
        rng = random.Random(seed * 131 + me)
        left, right = (me - 1 if me > 0 else None), (me + 1 if me < n_ranks - 1 else None)

        def hold(p, most):
            if me == stall_rank and rng.random() < 0.3:
                time.sleep(0.002)
            elif rng.random() < p:
                time.sleep(rng.random() * most)

        try:
            for t in range(1, steps + 1):
                par = t & 1
                # head (or the first workgroups of the density launch): push this step's update, raise the neighbours' flags ...
                for nb, side_there in ((left, 1), (right, 0)):
                    if nb is not None:
                        blocks[nb].recv[side_there][par] = ("update", t, me)
                        blocks[nb].flag[side_there] = 2 * t
                # ... wait for theirs ("at least": a neighbour may be a launch ahead), unpack
                for nb, side in ((left, 0), (right, 1)):
                    if nb is not None:
                        wait(lambda side=side: blocks[me].flag[side] >= 2 * t, "rank %d step %d: update flag, side %d" % (me, t, side))
                        got = blocks[me].recv[side][par]
                        if got != ("update", t, nb):
                            errors.append((me, t, side, got))
                hold(0.3, 0.0003)      # the density launch; a rank may be held up anywhere
                # gate: the MAX of the word over the ranks
                for q in range(n_ranks):
                    blocks[q].slots[par][me] = (t << 2) | words[me][t]
                for q in range(n_ranks):
                    wait(lambda q=q: blocks[me].slots[par][q] >> 2 == t, "rank %d step %d: word of rank %d" % (me, t, q))
                word = max(blocks[me].slots[par][q] & 3 for q in range(n_ranks))
                hold(0.3, 0.0003)
                if word != 0:      # rebuild: records to the neighbours (over the update of this parity), flags 2 t + 1, wait for mine, ingest
                    rebuilt[me].append(t)
                    for nb, side_there in ((left, 1), (right, 0)):
                        if nb is not None:
                            blocks[nb].recv[side_there][par] = ("records", t, me)
                            blocks[nb].flag[side_there] = 2 * t + 1
                    for nb, side in ((left, 0), (right, 1)):
                        if nb is not None:
                            wait(lambda side=side: blocks[me].flag[side] >= 2 * t + 1, "rank %d step %d: records flag, side %d" % (me, t, side))
                            got = blocks[me].recv[side][par]
                            if got != ("records", t, nb):
                                errors.append((me, t, side, got))
                hold(0.3, 0.0002)      # force
        except TimeoutError as e:
            errors.append(("timeout", str(e)))

    threads = [threading.Thread(target=rank, args=(r,)) for r in range(n_ranks)]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    return errors, rebuilt


@pytest.mark.parametrize("n_ranks", [2, 3, 4])
def test_speculative_step_every_rank_reads_the_message_of_its_own_step(n_ranks):
    for seed in range(3):
        errors, rebuilt = run_model(n_ranks, 240, True, seed, speculative=True)
        assert not errors, errors[:3]
        assert all(rb == rebuilt[0] for rb in rebuilt) and len(rebuilt[0]) > 10      # the same rebuild steps everywhere


@pytest.mark.parametrize("n_ranks", [2, 3, 4])
def test_every_rank_reads_the_message_of_its_own_step(n_ranks):
    for seed in range(3):
        errors, rebuilt = run_model(n_ranks, 240, True, seed)
        assert not errors, errors[:3]
        assert all(rb == rebuilt[0] for rb in rebuilt) and len(rebuilt[0]) > 10      # the same rebuild steps everywhere


def test_waiting_for_equality_gives_up_when_a_neighbour_is_a_launch_ahead():
    errors, _ = run_model(3, 120, False, 0, timeout=3.0)
    # (whichever flag wait is overtaken first: the update's, or — a neighbour that is through its rebuild and into the next head
    # kernel before this rank has looked — the records')
    assert any(e[0] == "timeout" and " flag, side" in e[1] for e in errors), errors[:3]
