"""Generates the assembly tile bodies of chain2.hip (chain2_*.inc): one 64-element tile = 32 ds_read_b128 (or 64
ds_read_b64) feeding 64 dependent v_add_f64, with the instruction order and the s_waitcnt counts written out, so the
question "can ONE wave overlap its LDS reads with its dependent adds?" is answered by the hardware, not by the compiler's
schedule. %0 = the accumulator (VGPR pair), %1 = the LDS byte address of the lane's row. usage: python gen_chain2.py"""
BASE = 20   # first ring register


def write(name, lines, regs):
    body = "\\n\\t".join(lines)
    clob = ",".join(f'"v{r}"' for r in regs)
    open(f"chain2_{name}.inc", "w").write(f'asm volatile("{body}" : "+v"(acc) : "v"(addr) : {clob}, "memory");\n')


def add_add_read(R):            # wait; add lo; add hi; refill the slot -- R reads in flight
    lines = [f"ds_read_b128 v[{BASE + 4 * k}:{BASE + 4 * k + 3}], %1 offset:{16 * k}" for k in range(R)]
    for k in range(32):
        q = BASE + 4 * (k % R)
        lines.append(f"s_waitcnt lgkmcnt({R - 1 if k + R <= 32 else 32 - k - 1})")
        lines.append(f"v_add_f64 %0, %0, v[{q}:{q + 1}]")
        lines.append(f"v_add_f64 %0, %0, v[{q + 2}:{q + 3}]")
        if k + R < 32:
            lines.append(f"ds_read_b128 v[{q}:{q + 3}], %1 offset:{16 * (k + R)}")
    return lines, range(BASE, BASE + 4 * R)


def add_read_add(R):            # the refill issued BETWEEN the two adds (hi half copied out first)
    lines = [f"ds_read_b128 v[{BASE + 4 * k}:{BASE + 4 * k + 3}], %1 offset:{16 * k}" for k in range(R)]
    for k in range(32):
        q = BASE + 4 * (k % R)
        lines.append(f"s_waitcnt lgkmcnt({R - 1 if k + R <= 32 else 32 - k - 1})")
        lines.append(f"v_add_f64 %0, %0, v[{q}:{q + 1}]")
        lines += [f"v_mov_b32 v18, v{q + 2}", f"v_mov_b32 v19, v{q + 3}"]
        if k + R < 32:
            lines.append(f"ds_read_b128 v[{q}:{q + 3}], %1 offset:{16 * (k + R)}")
        lines.append("v_add_f64 %0, %0, v[18:19]")
    return lines, range(18, BASE + 4 * R)


def b64(R):                     # 64 x ds_read_b64, one per add
    lines = [f"ds_read_b64 v[{BASE + 2 * k}:{BASE + 2 * k + 1}], %1 offset:{8 * k}" for k in range(R)]
    for k in range(64):
        q = BASE + 2 * (k % R)
        lines.append(f"s_waitcnt lgkmcnt({R - 1 if k + R <= 64 else 64 - k - 1})")
        lines.append(f"v_add_f64 %0, %0, v[{q}:{q + 1}]")
        if k + R < 64:
            lines.append(f"ds_read_b64 v[{q}:{q + 1}], %1 offset:{8 * (k + R)}")
    return lines, range(BASE, BASE + 2 * R)


write("r8", *add_add_read(8))
write("r4", *add_add_read(4))
write("r16", *add_add_read(16))
write("m8", *add_read_add(8))
write("b64", *b64(16))
write("adds", ["v_add_f64 %0, %0, v[20:21]"] * 64, [20, 21])
write("reads", [f"ds_read_b128 v[{BASE + 4 * (k % 8)}:{BASE + 4 * (k % 8) + 3}], %1 offset:{16 * k}" for k in range(32)] +
      ["s_waitcnt lgkmcnt(0)"], range(BASE, BASE + 32))


def global_ring(R):             # the same ring fed by global_load_dwordx4 (vmcnt) instead of ds_read_b128: do VMEM returns overlap the adds?
    lines = ["s_waitcnt vmcnt(0)"] + [f"global_load_dwordx4 v[{BASE + 4 * k}:{BASE + 4 * k + 3}], %1, off offset:{16 * k}" for k in range(R)]
    for k in range(32):
        q = BASE + 4 * (k % R)
        lines.append(f"s_waitcnt vmcnt({R - 1 if k + R <= 32 else 32 - k - 1})")
        lines.append(f"v_add_f64 %0, %0, v[{q}:{q + 1}]")
        lines.append(f"v_add_f64 %0, %0, v[{q + 2}:{q + 3}]")
        if k + R < 32:
            lines.append(f"global_load_dwordx4 v[{q}:{q + 3}], %1, off offset:{16 * (k + R)}")
    return lines, range(BASE, BASE + 4 * R)


write("g16", *global_ring(16))
write("greads", ["s_waitcnt vmcnt(0)"] + [f"global_load_dwordx4 v[{BASE + 4 * (k % 8)}:{BASE + 4 * (k % 8) + 3}], %1, off offset:{16 * k}" for k in range(32)] +
      ["s_waitcnt vmcnt(0)"], range(BASE, BASE + 32))
