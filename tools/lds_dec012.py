"""Developer tool (here): LDS bank-conflict model of dec012_mfma -- the B-fragment reads of the three blocks (ds_read_b128) AND the
scatter epilogues that write block j's output into block j+1's tile with that tile's swizzle (ds_write_b64), for the swizzle the
host picks today (choose_swz: reads only) and for a search over the family that prices both.  68x120 geometry.
MI355X_MICROARCH.md, LDS: ds_read_b128 = 4 groups of 16 lanes, bank of a 16-byte piece = (a / 16) mod 16; ds_write_b64 = 4 groups of
16 CONTIGUOUS lanes, bank (a / 4) mod 32, i.e. 8-byte slot (a / 8) mod 16; each extra distinct address on a busy slot = one cycle."""
import itertools
import sys

RG = [list(range(0, 4)) + list(range(12, 16)) + list(range(20, 28)),
      list(range(4, 12)) + list(range(16, 20)) + list(range(28, 32))]
RG = RG + [[32 + x for x in g] for g in RG]
WG = [list(range(16 * k, 16 * k + 16)) for k in range(4)]


def cyc(addrs, groups, slot):
    tot = 0
    for g in groups:
        slots = {}
        for l in g:
            if addrs[l] is None:
                continue
            slots.setdefault(slot(addrs[l]), set()).add(addrs[l])
        tot += max((len(v) for v in slots.values()), default=0)
    return tot


def swz_fn(p, a, b, c, L, cpp):
    return lambda xx, yy: ((((yy * L + xx) >> p) * a) + yy * b + (yy & 1) * c) % cpp


def reads(C, Hi, Wi, swz):
    TC, PS, KC, GW, GH = Wi + 2, C * 2, C // 16, Wi + 1, Hi + 1
    npos = GH * GW
    tot = ideal = 0
    for tile in range((npos + 31) // 32):
        for a, b, kc in itertools.product(range(2), range(2), range(KC)):
            addrs = []
            for lane in range(64):
                q = min(tile * 32 + (lane & 31), npos - 1)
                u, v = divmod(q, GW)
                yy, xx = u + 1 - a, v + 1 - b
                addrs.append((yy * TC + xx) * PS + ((kc * 2 + (lane >> 5)) ^ swz(xx, yy)) * 16)
            tot += cyc(addrs, RG, lambda x: (x // 16) % 16)
            ideal += 4
    return tot, ideal


def writes(COUT, Hi, Wi, Hd, Wd, cy, cx, CN, Wn, swz):
    """block (input Hi x Wi, COUT channels out, crop cy / cx) scattering into the next tile (CN channels per pixel, row of Wn + 2 pixels)."""
    GW, GH, MT = Wi + 1, Hi + 1, 4 * COUT // 32
    TCN, PSN = Wn + 2, CN * 2
    npos = GH * GW
    tot = ideal = 0
    for tile in range((npos + 31) // 32):
        for mtile in range(MT):
            for gq in range(4):
                addrs = []
                for lane in range(64):
                    kh = lane >> 5
                    q = tile * 32 + (lane & 31)
                    nb = mtile * 32 + 4 * kh
                    phase, cob = nb // COUT, nb % COUT
                    if q >= npos:
                        addrs.append(None)
                        continue
                    u, v = divmod(q, GW)
                    Y, X = 2 * u + (phase >> 1) - cy, 2 * v + (phase & 1) - cx
                    if not (0 <= Y < Hd and 0 <= X < Wd):
                        addrs.append(None)
                        continue
                    yy, xx = Y + 1, X + 1
                    addrs.append((yy * TCN + xx) * PSN + (cob & 7) * 2 + ((((cob >> 3) + gq) ^ swz(xx, yy)) * 16))
                tot += cyc(addrs, WG, lambda x: (x // 8) % 16)
                ideal += 4
    return tot, ideal


def family(cpp, TC):
    for L in (0, TC):
        for p in range(3):
            for a in sorted({0, 1 % cpp, 3 % cpp}):
                for b in range(cpp):
                    for c in range(cpp):
                        yield (p, a, b, c, L)


def host_choice(C, Hi, Wi):
    """choose_swz(false, C, Wi, 0, Hi + 1): first member with the fewest read cycles (stops at the ideal)."""
    cpp, best = C // 8, None
    for prm in family(cpp, Wi + 2):
        t, i = reads(C, Hi, Wi, swz_fn(*prm, cpp))
        if best is None or t < best[0]:
            best = (t, prm)
        if t == i:
            break
    return best[1]


if __name__ == "__main__":
    # level geometry at 68x120: lv[4] 4x7 (128 ch), lv[3] 9x15 (64), lv[2] 17x30 (32), lv[1] 34x60
    blocks = [dict(C=128, Hi=4, Wi=7, COUT=64, Hd=9, Wd=15), dict(C=128, Hi=9, Wi=15, COUT=32, Hd=17, Wd=30), dict(C=64, Hi=17, Wi=30, COUT=16, Hd=34, Wd=60)]
    crops = [(int(x.split(",")[0]), int(x.split(",")[1])) for x in (sys.argv[2:5] if len(sys.argv) > 4 else ["1,1", "1,1", "1,1"])]
    total = 0
    for j, bl in enumerate(blocks):
        prm = host_choice(bl["C"], bl["Hi"], bl["Wi"])
        r, ri = reads(bl["C"], bl["Hi"], bl["Wi"], swz_fn(*prm, bl["C"] // 8))
        line = f"tile {j} (C={bl['C']}, {bl['Hi']}x{bl['Wi']}): host swizzle {prm}: reads {r}/{ri}"
        extra = r - ri
        if j > 0:
            pb = blocks[j - 1]
            w, wi = writes(pb["COUT"], pb["Hi"], pb["Wi"], pb["Hd"], pb["Wd"], *crops[j - 1], bl["C"], bl["Wi"], swz_fn(*prm, bl["C"] // 8))
            line += f"; block {j-1}'s scatter writes {w}/{wi}"
            extra += w - wi
        print(line, f"-> {extra} extra cycles per frame")
        total += extra
    print("modelled extra LDS cycles per frame:", total, "(counter: 1,060,864 / 256 = 4,144)")
    if len(sys.argv) > 1 and sys.argv[1] == "search":
        for j in (1, 2):
            bl, pb = blocks[j], blocks[j - 1]
            cpp = bl["C"] // 8
            best = []
            for prm in family(cpp, bl["Wi"] + 2):
                f = swz_fn(*prm, cpp)
                r, ri = reads(bl["C"], bl["Hi"], bl["Wi"], f)
                w, wi = writes(pb["COUT"], pb["Hi"], pb["Wi"], pb["Hd"], pb["Wd"], *crops[j - 1], bl["C"], bl["Wi"], f)
                best.append((r - ri + w - wi, r - ri, w - wi, prm))
            best.sort()
            print(f"tile {j}: best members by reads + writes (extra total, reads, writes, (p, a, b, c, L)):", best[:5])
