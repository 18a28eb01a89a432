"""The schedule of zk_all_to_all_v (csrc/comm.hip a2a_plan, exported as zk_comm_plan) on the CPU, for the world sizes the GPU box cannot
run: who talks to whom in which round, at which byte offsets.  For a consistent set of per-rank counts (what parallel.Exchange passes:
recv_cnt[r] on rank q == send_cnt[q] on rank r) and every rank's plan:
  * every send (q -> p, round j, n bytes) meets exactly one receive on p from q in the same round with the same n -- RCCL pairs the
    sends and receives of a group by peer, so an unmatched or mis-sized one hangs or corrupts;
  * the sends of a rank to a peer cover that piece of its send buffer once, in order; likewise the receives;
  * a rank talks to (me + d) and (me - d) in the same step: both directions of every link in one group.
No GPU, no RCCL: the function is pure arithmetic on the counts.  (The one-rank GPU test, tests/_rccl_one_rank.py, executes the same
plan through ncclSend / ncclRecv to self.)"""
import numpy as np
import pytest

from zotmer_amd import native


def plans(world, cnt, eb, chunk, self_loop=False):
    """cnt[q][p] = elements rank q sends to rank p"""
    out = []
    for q in range(world):
        send_cnt = [cnt[q][p] for p in range(world)]
        recv_cnt = [cnt[p][q] for p in range(world)]
        send_off = np.concatenate([[0], np.cumsum(send_cnt)[:-1]]) + 3          # pieces back to back, the buffer starting at element 3
        recv_off = np.concatenate([[0], np.cumsum(recv_cnt)[:-1]]) + 5
        out.append((native.Context.comm_plan(world, q, send_off, send_cnt, recv_off, recv_cnt, eb, chunk, self_loop), send_off, recv_off))
    return out


@pytest.mark.parametrize("world", [1, 2, 3, 4, 7, 8])
@pytest.mark.parametrize("eb,chunk", [(8, 1000), (4, 4096), (8, 0), (12, 777)])
def test_every_send_meets_its_receive(world, eb, chunk):
    rng = np.random.default_rng(world * 100 + eb)
    cnt = rng.integers(0, 2000, size=(world, world))
    cnt[rng.random((world, world)) < 0.15] = 0          # empty pieces
    if world > 2:
        cnt[1][2] = 100_000                              # one piece of many rounds beside short ones
    for self_loop in (False, True):
        ps = plans(world, cnt, eb, chunk, self_loop)
        sends, recvs = {}, {}
        for q, (ops, send_off, recv_off) in enumerate(ps):
            covered_s = {p: [] for p in range(world)}
            covered_r = {p: [] for p in range(world)}
            last_round = -1
            for o in ops:
                assert o["round"] >= last_round, "rounds are issued in order"
                last_round = o["round"]
                assert o["bytes"] > 0 and (chunk == 0 or o["bytes"] <= chunk)
                key = (q, o["peer"], o["round"]) if not o["recv"] else (o["peer"], q, o["round"])
                d = recvs if o["recv"] else sends
                assert key not in d, "two messages between one pair in one round and direction"
                d[key] = o["bytes"]
                (covered_r if o["recv"] else covered_s)[o["peer"]].append((o["offset"], o["bytes"]))
                if not self_loop:
                    assert o["peer"] != q
            for p in range(world):
                for cov, off0, n in ((covered_s[p], int(send_off[p]) * eb, int(cnt[q][p]) * eb), (covered_r[p], int(recv_off[p]) * eb, int(cnt[p][q]) * eb)):
                    if p == q and not self_loop:
                        assert cov == []          # the kept piece is a device copy
                        continue
                    at = off0
                    for o_off, o_len in cov:
                        assert o_off == at, "pieces are covered in order without gaps"
                        at += o_len
                    assert at == off0 + n
        assert sends == recvs, "every send has its receive: same pair, same round, same length"


def test_both_directions_of_a_link_share_a_group():
    world, eb = 8, 8
    cnt = np.full((world, world), 1000)
    for q, (ops, _, _) in enumerate(plans(world, cnt, eb, 0)):
        # one round; within it the steps d = 1 .. 7: send to q + d, receive from q - d
        assert [o["round"] for o in ops] == [0] * (2 * (world - 1))
        for d in range(1, world):
            s, r = ops[2 * (d - 1)], ops[2 * (d - 1) + 1]
            assert (s["recv"], s["peer"]) == (0, (q + d) % world) and (r["recv"], r["peer"]) == (1, (q - d) % world)


def test_bad_arguments_are_refused():
    with pytest.raises(native.ZotkError):
        native.Context.comm_plan(2, 2, [0, 0], [1, 1], [0, 0], [1, 1], 8)
