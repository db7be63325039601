"""Host replay of the control flow of tri_inverse_cols4_kernel (csrc/gp_pretrain.hip): four waves share a block column of U^-1, meet at two
LDS-only barriers per step and each walks its own stream of prefetched items.  A wave that took one barrier more or less than the others would
hang the workgroup on the device, and a product issued before its block's barrier would read an unfinished block -- both are properties of the
scalar bookkeeping alone, so they are checked here, on the CPU, for every column length the kernel can meet (N <= 1152: J <= 71).

The replay mirrors the kernel line by line (skip_empty / fetch / end_products / idle_steps / the unrolled ring loop); keep the two in step."""
import pytest

PF = 4  # TC4_PF


def replay_wave(J, w):
    fetched = []
    If, mf = J - 1, w

    def skip_empty(I, m):
        while I >= 0 and m >= 0 and m >= J - I:
            if w == 0:
                return I, -1
            I -= 1
            m = w
        return I, m

    If, mf = skip_empty(If, mf)

    def fetch():
        nonlocal If, mf
        if If >= 0:
            fetched.append((If, mf))
            if mf < 0:
                If -= 1
                mf = w
            else:
                mf += 4
            If, mf = skip_empty(If, mf)
        else:
            fetched.append(None)
        return fetched[-1]

    ring = [fetch() for _ in range(PF)]
    barriers, work = ["init"], []
    Ip, mp, newest = J - 1, w, False

    def end_products():
        nonlocal newest
        if not newest:
            barriers.append(("B", Ip))
        barriers.append(("A", Ip))
        newest = False

    def idle_steps():
        nonlocal Ip, mp
        while Ip >= 0 and mp >= 0 and mp >= J - Ip:
            end_products()
            if w == 0:
                mp = -1
                return
            Ip -= 1
            mp = w

    idle_steps()
    rounds = 0
    while Ip >= 0:
        for q in range(PF):
            if Ip >= 0:
                assert ring[q] == (Ip, mp), "the ring holds another item than the one processed"
                if mp >= 0:
                    n, K = J - Ip, J - mp
                    if mp == n - 1:
                        barriers.append(("B", Ip))
                        newest = True
                    work.append(("product", Ip, K, len(barriers)))
                    mp += 4
                    if mp >= n:
                        end_products()
                        if w == 0:
                            mp = -1
                        else:
                            Ip -= 1
                            mp = w
                            idle_steps()
                else:
                    work.append(("close", Ip, None, len(barriers)))
                    Ip -= 1
                    mp = 0
                    idle_steps()
            ring[q] = fetch()
        rounds += 1
        assert rounds < 100000
    barriers.append("final")
    return barriers, work


@pytest.mark.parametrize("J", list(range(0, 30)) + [47, 48, 71])
def test_four_wave_inverse_column_schedule(J):
    waves = [replay_wave(J, w) for w in range(4)]
    seq = waves[0][0]
    assert all(b == seq for b, _ in waves), "the waves do not take the same barriers"
    assert len(seq) == 2 + 2 * J  # B and A per step, one before, one after
    for _, work in waves:
        for kind, I, K, passed in work:
            if kind == "product" and K < J:  # block K was closed in step K: visible after barrier B of step K - 1
                assert seq.index(("B", K - 1)) < passed
            if kind == "close":  # the partial sums of step I: after its barrier A
                assert seq.index(("A", I)) < passed
    products = sorted((I, K) for _, work in waves for kind, I, K, _ in work if kind == "product")
    assert products == [(I, K) for I in range(J) for K in range(I + 1, J + 1)]
    closes = [I for kind, I, _, _ in waves[0][1] if kind == "close"]
    assert closes == list(range(J - 1, -1, -1)) and not any(kind == "close" for _, work in waves[1:] for kind, *_ in work)
