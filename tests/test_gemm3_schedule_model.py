"""A model of the eight-phase GEMM schedules of csrc/gemm3.hip (round 5) that checks their hazard rules on the CPU.

The kernels order LDS-DMA writes and ds_reads of a two-buffer ring by COUNTED s_waitcnt vmcnt + s_barrier only, with two wave groups
running the same phase sequence one barrier apart.  The file header derives two rules from the barrier numbering (group 0's phase g
reads / stages in (#2g-1, #2g) and multiplies in (#2g, #2g+1); group 1's in (#2g, #2g+1) and (#2g+1, #2g+2)):

  RAW  a half-tile may be read in phase r only if some phase w <= r - 1 carries a counted wait that retires it (every wave waits in
       front of phase w's first barrier for its own pieces; everybody's are visible once barrier #2w+1 has opened, i.e. from phase w + 1);
  WAR  a half-tile's slot may be restaged in phase p only if p >= (phase of its last read) + 1 (every read is retired -- lgkmcnt(0) --
       in front of the reading phase's first barrier).

This test restates each kernel's per-phase actions (the K3_SLICE / G3_SLICE macros and the prologues) as tables and replays them
for many slices: every read must find its half-tile retired by the in-order load count, every restage must come after the slot's last
read, and the loads in flight must fit the 6-bit vmcnt.  It does not parse the source: the tables below ARE the documented schedule;
tests/test_gemm_k3_gpu.py checks the kernels against another kernel family to the bit on hardware."""
import pytest

# one phase: (reads, stage, vmcnt) -- reads / stage name half-tiles as (operand-half, slice offset from the phase's slice); a stage's
# load count per wave is LOADS[operand]; vmcnt = the counted wait after the phase's staging (None: no wait in this phase)
K3_256x256 = dict(
    loads={"A": 2, "B": 2},
    prologue=[("B0", 0), ("A0", 0), ("B1", 0), ("A1", 0), ("B0", 1), ("A0", 1), ("B1", 1), ("A1", 1)],
    prologue_vmcnt=12, prologue_reads=[("B0", 0)],
    phases=[([("A0", 0)], ("B0", 2), 12),
            ([("B1", 0)], ("A0", 2), 12),
            ([("A1", 0)], ("B1", 2), 12),
            ([("B0", 1)], ("A1", 2), 12)])
K3_256x320_GEGLU = dict(
    loads={"A": 2, "B": 3},
    prologue=[("A0", 0), ("B0", 0), ("A1", 0), ("B1", 0), ("B0", 1), ("A1", 1), ("B1", 1)],
    prologue_vmcnt=8, prologue_reads=[],
    phases=[([("A0", 0), ("B0", 0)], ("A0", 1), None),
            ([("A1", 0)], ("B0", 2), None),
            ([("B1", 0)], ("A1", 2), None),
            ([("A0", 0)], ("B1", 2), 8)])


def replay(sched, n_slices):
    loads = sched["loads"]
    issued = []            # per wave, in issue order: (half-tile name, slice, first load index, load count)
    n_issued = 0
    retired_upto = 0       # loads [0, retired_upto) have been retired by a counted wait ...
    retired_phase = {}     # ... and (name, slice) -> the phase whose wait retired it
    last_read = {}         # slot (name, slice parity) -> phase of the last read of what it holds
    holds = {}             # slot -> slice it holds (or is being filled with)
    max_in_flight = 0

    def stage(name, s, phase):
        nonlocal n_issued, max_in_flight
        slot = (name, s & 1)
        if slot in holds:
            prev = holds[slot]
            assert prev == s - 2, f"{name} of slice {s} overwrites slice {prev}"
            lr = last_read.get(slot)
            assert lr is not None, f"phase {phase}: {name}({prev}) is overwritten without ever being read"
            assert phase >= lr + 1, f"WAR: phase {phase} restages {name}'s slot, last read in phase {lr}"
        holds[slot] = s
        last_read.pop(slot, None)
        issued.append((name, s, n_issued, loads[name[0]]))
        n_issued += loads[name[0]]
        max_in_flight = max(max_in_flight, n_issued - retired_upto)

    def wait(vmcnt, phase):
        nonlocal retired_upto
        upto = max(retired_upto, n_issued - vmcnt)
        for name, s, first, cnt in issued:
            if first + cnt <= upto and (name, s) not in retired_phase:
                retired_phase[(name, s)] = phase
        retired_upto = upto

    def read(name, s, phase):
        slot = (name, s & 1)
        assert holds.get(slot) == s, f"phase {phase}: reads {name}({s}) but the slot holds slice {holds.get(slot)}"
        w = retired_phase.get((name, s))
        assert w is not None and w <= phase - 1, f"RAW: phase {phase} reads {name}({s}), retired by the wait of phase {w}"
        last_read[slot] = phase

    for name, s in sched["prologue"]:
        stage(name, s, -2)
    wait(sched["prologue_vmcnt"], -2)
    for name, s in sched["prologue_reads"]:
        read(name, s, -1)
    for s in range(n_slices):
        for k, (reads, (sname, soff), vmcnt) in enumerate(sched["phases"]):
            phase = 4 * s + k
            for name, off in reads:
                read(name, s + off, phase)
            stage(sname, s + soff, phase)
            if vmcnt is not None:
                wait(vmcnt, phase)
    return max_in_flight


@pytest.mark.parametrize("sched", [K3_256x256, K3_256x320_GEGLU], ids=["k3:256x256", "k3:256x320"])
def test_eight_phase_schedule_orders_every_read_and_restage(sched):
    in_flight = replay(sched, 40)
    assert in_flight <= 63, "vmcnt is a 6-bit counter"
    # wave 0 also carries ONE older load (the bias piece): it is the first to retire, so every count above holds with it in front


def test_the_model_catches_what_the_rules_forbid():
    """The checker is not vacuous: the two schedules that were built and discarded for hazards it must flag, it flags."""
    import copy
    early = copy.deepcopy(K3_256x256)                      # B0 of the next slice read one phase early (P3 instead of P4)
    early["phases"][2] = ([("A1", 0), ("B0", 1)], ("B1", 2), 12)
    early["phases"][3] = ([], ("A1", 2), 12)
    with pytest.raises(AssertionError, match="RAW"):
        replay(early, 8)
    eager = copy.deepcopy(K3_256x320_GEGLU)                # A0 restaged in the phase that reads it for the second time
    eager["phases"][3] = ([("A0", 0)], ("A0", 1), 8)
    eager["phases"][0] = ([("A0", 0), ("B0", 0)], ("B1", 2), None)
    with pytest.raises(AssertionError):
        replay(eager, 8)
    shallow = copy.deepcopy(K3_256x320_GEGLU)              # a wait that leaves one more half-tile in flight than the reads allow
    shallow["phases"][3] = ([("A0", 0)], ("B1", 2), 10)
    with pytest.raises(AssertionError, match="RAW"):
        replay(shallow, 8)
