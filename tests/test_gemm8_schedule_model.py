"""The hazard rules of the 8-phase K loop's BALANCED DMA schedules (cellvit_amd/csrc/gemm8.hip: fp16 linear / qkv and fp8), checked on the schedule the SOURCE TEXT
spells out: the loop bodies and prologues are parsed (stage_a / stage_w / stage_sc calls with their K tile and piece ranges, fragment reads, counted vmcnt waits, the
`more` / `dm` branches), replayed for K loops of 1 .. 5 iterations, and every event is checked against the rules in the kernel's header:
  WAR  a sub-tile is re-staged >= 2 phases after its last fragment read (the two wave groups run one barrier = half a phase apart);
  RAW  a sub-tile is read >= 1 phase after a counted wait that retires ALL of its pieces — vmcnt(n) retires everything but the n most recently issued
       operations, loads retire in order (stores in between only make a wait stricter, so they are left out: the worst case);
  and the K tile a read finds in its buffer is the one the MFMAs of that phase expect.
A wrong count in one of those waits is a race that a GPU test may never hit; this is a CPU test."""
import os
import re

import pytest

SRC = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "cellvit_amd", "csrc", "gemm8.hip")


def _strip(text):
    return re.sub(r"//[^\n]*", "", text)


_IF = re.compile(r"\s*if\s*(constexpr\s*)?\(")


def _stmt_end(text, j):
    """Index just behind the statement that starts at j: `{ ... }`, `if (...) stmt [else stmt]`, or `...;`."""
    n = len(text)
    while j < n and text[j].isspace():
        j += 1
    if text[j] == "{":
        depth, k = 0, j
        while True:
            depth += text[k] == "{"
            depth -= text[k] == "}"
            k += 1
            if depth == 0:
                return k
    m = _IF.match(text, j)
    if m:
        depth, k = 1, m.end()
        while depth:
            depth += text[k] == "("
            depth -= text[k] == ")"
            k += 1
        k = _stmt_end(text, k)
        m2 = re.compile(r"\s*else\b").match(text, k)
        return _stmt_end(text, m2.end()) if m2 else k
    return text.index(";", j) + 1


def _run(text, env):
    """Primitive statements executed by `text` under the boolean environment `env` (names, !names, a && b; anything unknown is false)."""
    out, i, n = [], 0, len(text)

    def cond(c):
        c = c.strip()
        if "&&" in c:
            return all(cond(x) for x in c.split("&&"))
        return (not env.get(c[1:].strip(), False)) if c.startswith("!") else env.get(c, False)

    while i < n:
        if text[i].isspace():
            i += 1
            continue
        if text[i] == "{":
            k = _stmt_end(text, i)
            out += _run(text[i + 1:k - 1], env)
            i = k
            continue
        m = _IF.match(text, i)
        if m:
            depth, k = 1, m.end()
            while depth:
                depth += text[k] == "("
                depth -= text[k] == ")"
                k += 1
            c = text[m.end():k - 1]
            k2 = _stmt_end(text, k)
            then = text[k:k2]
            m2 = re.compile(r"\s*else\b").match(text, k2)
            other, k3 = None, k2
            if m2:
                k3 = _stmt_end(text, m2.end())
                other = text[m2.end():k3]
            if cond(c):
                out += _run(then, env)
            elif other is not None:
                out += _run(other, env)
            i = k3
            continue
        k = text.index(";", i)
        out.append(text[i:k].strip())
        i = k + 1
    return [s for s in out if s]


def _events(stmts, fp8):
    ev = []
    for s in stmts:
        m = re.fullmatch(r"stage_(a|w)\((\d), (?:kt \+ )?(\d)(?:, (\d), (\d))?\)", s)
        if m:
            ev.append(("dma", m.group(1).upper(), int(m.group(2)), int(m.group(3)), int(m.group(4) or 0), int(m.group(5) or 4)))
            continue
        m = re.fullmatch(r"stage_sc\((\d), (?:kt \+ )?(\d)\)", s)
        if m:
            ev.append(("sc", int(m.group(1)), int(m.group(2))))
            continue
        m = re.fullmatch(r"G8_VMCNT\((\d+)\)", s)
        if m:
            ev.append(("wait", int(m.group(1))))
            continue
        m = re.fullmatch(r"G8_RD_(A|W)\(\w+, (\d), (\d)\)", s)
        if m:
            ev.append(("read", m.group(1), int(m.group(2)), int(m.group(3))))
            continue
        m = re.fullmatch(r"G8_RD_A_(LO|HI)\(\w+, (\d), (\d)\)", s)       # one half of an A sub-tile's fragments: the same sub-tile as far as the rules go
        if m:
            ev.append(("read", "A", int(m.group(2)), int(m.group(3))))
            continue
        m = re.fullmatch(r"G8F_RD_A\(\w+, \w+, (\d), (\d)\)", s)
        if m:
            ev += [("read", "A", int(m.group(1)), int(m.group(2))), ("read", "S", int(m.group(1)), 0)]
            continue
        m = re.fullmatch(r"G8F_RD_W\(\w+, (\d), (\d)\)", s)
        if m:
            ev.append(("read", "W", int(m.group(1)), int(m.group(2))))
            continue
        m = re.fullmatch(r"G8F_RD_SW\(\w+, (\d)\)", s)
        if m:
            ev.append(("read", "S", int(m.group(1)), 0))
            continue
        assert re.match(r"G8_(BAR8?|MMQ8?|PSTAMP|STAMP)\(|const bool (more|dm)\b", s), f"unparsed statement in the K loop: {s!r}"
    return ev


def _block(src, at):
    """The brace block whose `{` is the first one at or after index `at`: its inner text."""
    j = src.index("{", at)
    depth, k = 0, j
    while True:
        depth += src[k] == "{"
        depth -= src[k] == "}"
        k += 1
        if depth == 0:
            return src[j + 1:k - 1]


def _schedule(fp8, whole=False):
    """whole: the UNBALANCED fp16 schedule of the implicit-GEMM convolutions (a wave stages 32 contiguous rows: whole tiles per call)."""
    src = _strip(open(SRC).read())
    pro = _block(src, src.index("auto stage_prologue = "))
    prologue = _events([s for s in _run(pro, {"BAL": not whole, "F8": fp8}) if s.startswith("stage_")], fp8)
    f8 = _block(src, src.index("if constexpr (F8) {"))                          # the fp8 branch of the tile loop
    if whole:
        rest = src[src.index("if constexpr (F8) {") + len(f8):]
        skip = _block(rest, rest.index("if constexpr (BAL) {"))
        rest = rest[rest.index("if constexpr (BAL) {") + len(skip):]
        bal = _block(rest, rest.index("else"))
        top = _events(_run(bal[:bal.index("for (int kt")], {"no_rd": False, "no_dma": False, "wr == 1": False}), fp8)
    elif fp8:
        bal = _block(f8, f8.index("if constexpr (BAL) {"))
        top = _events(_run(f8[f8.index("G8_VMCNT"):f8.index("if constexpr (BAL) {")], {}), fp8)
    else:
        rest = src[src.index("if constexpr (F8) {") + len(f8):]
        bal = _block(rest, rest.index("if constexpr (BAL) {"))
        top = _events(_run(bal[:bal.index("for (int kt")], {"no_rd": False, "no_dma": False, "wr == 1": False}), fp8)
    loop = _block(bal, bal.index("for (int kt"))
    phases = re.split(r"G8_BAR8?\(\); G8_MMQ8?\([^;]*\); G8_BAR8?\(\);", loop)[:8]
    assert len(phases) == 8
    per = {more: [_events(_run(ph, {"more": more, "dm": more, "no_rd": False, "no_dma": False}), fp8) for ph in phases] for more in (True, False)}
    return prologue, top, per


def _replay(fp8, iters, whole=False):
    """Replays one output tile with `iters` iterations (2 * iters K tiles); returns the list of rule violations."""
    prologue, top, per = _schedule(fp8, whole)
    bad = []
    seq = 0
    piece = {}          # (kind, buf, i) -> (seq, ktile, phase issued); scale blocks: ('S', buf, 0)
    last_read = {}      # (kind, buf, sub) -> phase of the last read
    retired = -1        # every operation with seq <= retired has landed (by a wait), as of `retired_at`
    waits = []          # (phase, retired-up-to seq)

    def issue(kind, buf, i, ktile, t):
        nonlocal seq
        for sub in ((0, 1) if whole else ((i >> 1 if kind != "S" else 0),)):      # whole: some wave's piece i lies in either sub-tile
            lr = last_read.get((kind, buf, sub))
            if lr is not None and t - lr < 2:
                bad.append(f"WAR: {kind}{buf} sub {sub} re-staged in phase {t}, last read in phase {lr}")
        piece[(kind, buf, i)] = (seq, ktile, t)
        seq += 1

    def read(kind, buf, sub, expect, t):
        last_read[(kind, buf, sub)] = t
        if expect is None:
            return
        for i in ((0,) if kind == "S" else ((0, 1, 2, 3) if whole else (2 * sub, 2 * sub + 1))):
            if (kind, buf, i) not in piece:
                bad.append(f"RAW: {kind}{buf} piece {i} read in phase {t} was never staged")
                continue
            s_, kt_, _ = piece[(kind, buf, i)]
            if kt_ != expect:
                bad.append(f"content: {kind}{buf} piece {i} holds K tile {kt_} in phase {t}, the MFMAs expect {expect}")
            if not any(wt <= t - 1 and upto >= s_ for wt, upto in waits):
                bad.append(f"RAW: {kind}{buf} piece {i} (K tile {kt_}) read in phase {t} without a retiring wait one phase earlier")

    def apply(events, t, kt):
        for e in events:
            if e[0] == "dma":
                for i in range(e[4], e[5]):
                    issue(e[1], e[2], i, (kt + e[3]) if kt is not None else e[3], t)
            elif e[0] == "sc":
                issue("S", e[1], 0, (kt + e[2]) if kt is not None else e[2], t)
            elif e[0] == "wait":
                waits.append((t, seq - 1 - e[1]))
            else:
                kind, buf, sub = e[1], e[2], e[3]
                if kt is None:
                    expect = buf                                  # before the loop: E = K tile 0
                else:
                    expect = kt + buf
                    if not fp8 and buf == 0 and (t % 8) in (6, 7):    # fp16 phases 7 / 8: the NEXT iteration's first operands
                        expect = kt + 2 if kt + 2 < 2 * iters else None
                read(kind, buf, sub, expect, t)

    apply(prologue, -10, None)          # (long before: the previous tile's epilogue lies in between)
    apply([e for e in top if e[0] == "wait"], -2, None)     # tile top: wait, barrier (the wave groups are aligned here), then the first reads
    apply([e for e in top if e[0] != "wait"], -1, None)
    for it in range(iters):
        more = it + 1 < iters
        for ph in range(8):
            apply(per[more][ph], 8 * it + ph, 2 * it)
    return bad


@pytest.mark.parametrize("fp8", [False, True])
def test_schedule_is_parsed_completely(fp8):
    prologue, top, per = _schedule(fp8)
    n_pro = sum(e[5] - e[4] for e in prologue if e[0] == "dma") + sum(1 for e in prologue if e[0] == "sc")
    assert n_pro == (13 if fp8 else 14)                                        # E whole (+ scales) and the part of O the first phases do not stage
    per_iter = sum((e[5] - e[4]) if e[0] == "dma" else 1 for ph in per[True] for e in ph if e[0] in ("dma", "sc"))
    assert per_iter == (18 if fp8 else 16)                                     # 2 K tiles x (4 A + 4 W pieces) per wave (+ 2 scale pieces)
    assert max(sum((e[5] - e[4]) if e[0] == "dma" else 1 for e in ph if e[0] in ("dma", "sc")) for ph in per[True]) == 3     # balanced
    reads = sum(1 for ph in per[True] for e in ph if e[0] == "read" and e[1] != "S")
    assert reads == (8 if fp8 else 10)                                         # every sub-tile of E and O once per iteration (fp16: the A sub-tiles 0 in two halves)
    assert [e for e in top if e[0] == "wait"], "the tile-top wait"


@pytest.mark.parametrize("fp8", [False, True])
@pytest.mark.parametrize("iters", [1, 2, 3, 5])
def test_hazard_rules_hold(fp8, iters):
    assert _replay(fp8, iters) == []


@pytest.mark.parametrize("iters", [1, 2, 3, 5])
def test_hazard_rules_hold_for_the_convolutions_schedule(iters):
    """The four-pieces-in-phases-3-4-7-8 schedule the implicit-GEMM convolutions keep (whole tiles per staging call)."""
    prologue, top, per = _schedule(False, whole=True)
    assert sum(e[5] - e[4] for e in prologue if e[0] == "dma") == 16
    assert sorted(sum(e[5] - e[4] for e in ph if e[0] == "dma") for ph in per[True]) == [0, 0, 0, 0, 4, 4, 4, 4]
    assert _replay(False, iters, whole=True) == []


@pytest.mark.parametrize("old,new,rule", [
    ("stage_w(0, kt + 2, 0, 3); G8_VMCNT(9);", "stage_w(0, kt + 2, 0, 3); G8_VMCNT(11);", "RAW"),        # a wait loosened by two operations
    ("stage_a(0, kt + 2, 0, 2); G8_VMCNT(8);", "stage_a(0, kt + 2, 2, 4); G8_VMCNT(8);", "WAR"),         # E.A sub-tile 1 re-staged in the phase that reads it
    ("stage_a(1, kt + 3, 0, 2); G8_VMCNT(8);", "stage_a(1, kt + 3, 0, 2); G8_VMCNT(10);", "RAW"),        # the wait in front of the early half-read of E.A loosened
])
def test_the_model_sees_a_broken_schedule(monkeypatch, old, new, rule):
    """Sanity of the checker itself: break the fp16 loop's text and the corresponding rule must trip."""
    import builtins
    real_open = builtins.open
    text = real_open(SRC).read()
    assert text.count(old) == 1

    class _F:
        def __init__(self, s): self.s = s
        def read(self): return self.s
        def __enter__(self): return self
        def __exit__(self, *a): return False

    def fake_open(path, *a, **k):
        if os.path.abspath(str(path)) == SRC:
            return _F(text.replace(old, new))
        return real_open(path, *a, **k)
    monkeypatch.setattr(builtins, "open", fake_open)
    assert any(b.startswith(rule) for b in _replay(False, 3))


# ------------------------------------------------------------------------------------------------------------------------------------------
# The counted LDS waits of the fp16 loops: G8_MMQ(n, a, b, ..) = s_waitcnt lgkmcnt(n) in front of the 16 MFMAs on register sets a, b.  LDS reads return in
# order, so the wait retires everything but the n newest ds_reads: every read INTO a or b must be older than that.  (G8_RD_A = 8 reads, G8_RD_A_LO / _HI and G8_RD_W = 4.)
def _lds_waits(whole):
    src = _strip(open(SRC).read())
    f8 = _block(src, src.index("if constexpr (F8) {"))
    rest = src[src.index("if constexpr (F8) {") + len(f8):]
    bal = _block(rest, rest.index("if constexpr (BAL) {"))
    if whole:
        rest = rest[rest.index("if constexpr (BAL) {") + len(bal):]
        bal = _block(rest, rest.index("else"))
    pre = _run(bal[:bal.index("for (int kt")], {"no_rd": False, "no_dma": False})
    loop = _run(_block(bal, bal.index("for (int kt")), {"more": True, "dm": True, "no_rd": False, "no_dma": False})
    return pre, loop


@pytest.mark.parametrize("whole", [False, True])
def test_counted_lds_waits_cover_the_fragments_they_release(whole):
    pre, loop = _lds_waits(whole)
    seq, last = 0, {}                                   # register set -> sequence number of the newest read into it
    checked = 0
    for it, stmts in enumerate([pre, loop, loop, loop]):
        for st in stmts:
            m = re.fullmatch(r"G8_RD_(A|W)\((\w+), \d, \d\)", st)
            if m:
                seq += 8 if m.group(1) == "A" else 4
                last[m.group(2)] = seq
                continue
            m = re.fullmatch(r"G8_RD_A_(LO|HI)\((\w+), \d, \d\)", st)
            if m:
                seq += 4
                last[m.group(2)] = seq
                continue
            m = re.fullmatch(r"G8_MMQ\((\d+), (\w+), (\w+), \d, \d\)", st)
            if m:
                n = int(m.group(1))
                for reg in (m.group(2), m.group(3)):
                    assert reg in last, f"{reg} consumed before any read"
                    assert seq - last[reg] >= n, f"iteration {it}: MFMAs on {reg} behind lgkmcnt({n}), but its reads are among the {n} newest"
                assert seq - max(last[m.group(2)], last[m.group(3)]) == n, "the wait is tighter than it needs to be"
                checked += 1
    assert checked == 24


# ------------------------------------------------------------------------------------------------------------------------------------------
# Which rows a wave stages (a_piece_row / w_piece_row) and where its pieces land in LDS (one scalar base per call + immediates): evaluated from the source text.
def _piece_rows(kind):
    src = _strip(open(SRC).read())
    body = _block(src, src.index(f"auto {kind}_piece_row = "))
    plain = re.search(r"if \(!BAL\) return ([^;]+);", body).group(1)
    q = re.search(r"const int q = ([^;]+);", body).group(1)
    bal = re.search(r"return ([^;]+);\s*$", body.strip()).group(1)
    imm = re.search(r"G8_DMA\(%s_voff\[i\], base, dst \+ ([^;]+)\);" % kind, src).group(1)           # e.g. (i & 1) * 1024 + (i >> 1) * (BAL ? 8192 : 2048)
    imm = re.sub(r"\(BAL \? (\d+) : (\d+)\)", r"(\1 if BAL else \2)", imm)

    def row(wave, i, BAL):
        return eval(bal, {}, {"q": eval(q, {}, {"wave": wave, "i": i}), "i": i}) if BAL else eval(plain, {}, {"wave": wave, "i": i})

    def dst_off(i, BAL):
        return eval(imm, {}, {"i": i, "BAL": BAL})
    return row, dst_off


@pytest.mark.parametrize("kind,sub_rows", [("a", 64), ("w", 32)])
@pytest.mark.parametrize("BAL", [True, False])
def test_every_row_of_a_tile_is_staged_once_and_pieces_sit_where_the_immediates_say(kind, sub_rows, BAL):
    row, dst_off = _piece_rows(kind)
    covered = sorted(r for w in range(8) for i in range(4) for r in range(row(w, i, BAL), row(w, i, BAL) + 8))
    assert covered == list(range(256))                                       # 8 waves x 4 pieces x 8 rows = the 256-row tile, no row twice
    for w in range(8):
        for i in range(4):
            assert row(w, i, BAL) * 128 == row(w, 0, BAL) * 128 + dst_off(i, BAL)        # LDS address of piece i = the call's base + immediate
            if BAL:                                                          # pieces 0, 1 in the sub-tile the fragment reads call 0, pieces 2, 3 in sub-tile 1
                assert (row(w, i, BAL) // sub_rows) % 2 == i >> 1


def test_halo_convolution_filter_row_permutation():
    """conv.hip, cout_of_row: LDS row j*16 + 4g + r of the staged filter block carries the output channel that makes lane (g, li)'s 16 accumulator values of a
    pixel two runs of 8 consecutive channels, [8g, 8g + 8) and [32 + 8g, 32 + 8g + 8) — what the direct stores and the fused head's MFMA operands rely on."""
    src = _strip(open(os.path.join(os.path.dirname(SRC), "conv.hip")).read())
    expr = re.search(r"int cout_of_row\(int n\) \{ return ([^;]+); \}", src).group(1)
    cout = [eval(expr, {}, {"n": n}) for n in range(64)]
    assert sorted(cout) == list(range(64))
    for g in range(4):
        for q in range(2):
            run = [cout[(2 * q + h) * 16 + 4 * g + r] for h in range(2) for r in range(4)]       # fragments j = 2q, 2q + 1; the lane's rows 4g .. 4g + 3
            assert run == list(range(q * 32 + g * 8, q * 32 + g * 8 + 8))
