"""CPU: model of the slot rings of stage 1 of the two-stage Hessenberg-triangular reduction (csrc/ht_twostage.hip,
`epoch_record` / `epoch_wait`).  A producer stream fills slot L mod RING in step L and records `ready`; consumer streams
wait for `ready`, read the slot, and tell the producer that slots are free again only once per EPOCH = RING / 2 steps,
through two alternating events (the host is the bottleneck of that loop: every runtime call counts).  HIP semantics:
a wait refers to the most recent record of the event that was ENQUEUED before it.  Under every interleaving of the
streams a consumer must find in a slot what its step put there.
Round 6: the producer of both rings is the chain's stream itself (the panel QR of the next step is the second workgroup
of a step's RQ launch) and the consumer `sq` gets a step's hand-over -- the `ready` record and its own wait and launch --
one to ZBATCH = 3 steps LATE, in batches, when a later factorisation is launched (`flush_z`): `defer` below.
The epoch waits of the producer come BEFORE the flush of a step (the host evaluates the next panel's slot first)."""
import random

import pytest

RING, EPOCH = 8, 4


def enqueue(nsteps, consumers, defer=0):
    """the host loop: per-stream op lists; ops are ("wait", record_id | None), ("write", L), ("read", L), ("record", id)"""
    streams = {"P": []}
    for c in consumers:
        streams[c] = []
    last = {}                       # event name -> id of its most recent enqueued record
    next_id = [0]

    def record(stream, ev):
        next_id[0] += 1
        last[ev] = next_id[0]
        streams[stream].append(("record", next_id[0]))

    def wait(stream, ev):
        streams[stream].append(("wait", last.get(ev)))

    pending = []

    def hand_over(L):
        """defer = False: at once; defer = Z >= 1: the steps wait until Z of them are pending at the START of a later
        step (flush_z): one record -- of the last one -- and one wait per consumer for the whole batch"""
        if defer:
            pending.append(L)
            return
        flush([L])

    def flush(batch):
        record("P", ("ready", batch[-1] % RING))
        for c in consumers:
            wait(c, ("ready", batch[-1] % RING))
            for M in batch:
                streams[c].append(("read", M))
                if M % EPOCH == EPOCH - 1:                   # epoch_record(used_*, M, consumer)
                    record(c, (c, (M // EPOCH) % 2))

    for L in range(nsteps):
        if L >= RING and L % EPOCH == 0:                     # epoch_wait(used_*, L, producer)
            for c in consumers:
                wait("P", (c, (L // EPOCH) % 2))
        if defer and len(pending) >= defer:                  # flush_z(false): before the launch that writes step L's slot
            flush(list(pending)); pending.clear()
        streams["P"].append(("write", L))
        hand_over(L)
    if pending:
        flush(list(pending)); pending.clear()
    return streams


def run(streams, rng):
    done, slots, pos = set(), {}, {k: 0 for k in streams}
    while any(pos[k] < len(v) for k, v in streams.items()):
        ready = []
        for k, ops in streams.items():
            if pos[k] < len(ops):
                op, arg = ops[pos[k]]
                if op != "wait" or arg is None or arg in done:
                    ready.append(k)
        assert ready, "deadlock"
        k = rng.choice(ready)
        op, arg = streams[k][pos[k]]
        if op == "write":
            slots[arg % RING] = arg
        elif op == "read":
            assert slots.get(arg % RING) == arg, (k, arg, slots.get(arg % RING))
        elif op == "record":
            done.add(arg)
        pos[k] += 1


@pytest.mark.parametrize("defer", [0, 1, 3, 4])
@pytest.mark.parametrize("consumers", [("s",), ("s", "sq")])
def test_ring_slots_are_never_overwritten_before_they_are_read(consumers, defer):
    rng = random.Random(5)
    for trial in range(300):
        run(enqueue(rng.choice([1, 7, 8, 9, 23, 64]), consumers, defer), rng)


def test_the_model_notices_a_missing_wait():
    """the same loop with the producer's waits one epoch late must fail under some interleaving"""
    streams = enqueue(40, ("s",))
    streams["P"] = [op for op in streams["P"] if op[0] != "wait"]
    rng = random.Random(1)
    with pytest.raises(AssertionError):
        for trial in range(300):
            run({k: list(v) for k, v in streams.items()}, rng)


def test_the_model_notices_a_batch_that_is_too_long():
    """hand-overs in batches of more than RING / 2 steps let the producer overwrite a slot `sq` has not read (the
    library's ZBATCH = 3 is below the limit the model finds: 4 passes, 5 does not)"""
    rng = random.Random(3)
    with pytest.raises(AssertionError):
        for trial in range(100):
            run(enqueue(64, ("s", "sq"), 5), rng)
