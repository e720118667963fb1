// kas_order_relax.h — order kernel, RELAXATION form: P5 (computePreferenceLists, KAS:202-239 with the
// PreferenceListOrderTracker of KAS:244-302) for lists up to 3 wide, one wavefront per scenario.
// Included at the end of kas_solver_body.h (namespace kas, the helpers of that file).
//
// What P5 is: rows in ascending order; row p reads count[n][0], count[n][1] of its own holders, picks (least count,
// first in the rotated visit order wins ties), then adds 1 to count[list[r]][r] for every list position.  Row p
// therefore depends on the earlier rows that hold one of its nodes and on nothing else.
//
// The form.  64 consecutive rows are one tile, lane i evaluates row i of the tile, and every earlier tile is
// finished when a tile starts.  Inside the tile the sequential answer is the fixed point of
//     outcome(i) = picks( committed counts + what the rows j < i of the tile that hold the same node add )
// and "what the earlier rows add" is obtained from the LDS itself: an LDS atomic add with return gives every lane
// the word before the instruction plus the addends of the LOWER lanes of that instruction that named the same word
// — the LDS serves the lanes of one instruction in ascending lane order (kas_wave.h, lds_add_rtn_u32: measured, and
// re-checked by kas_ctx_create's self-test).  For that to be a prefix over ROWS, the tile's 192 (row, cell) pairs
// are added in row-major order: pair v = 3 row + cell is lane v mod 64 of instruction v div 64, three instructions
// per tile, so that an earlier row's pair is an earlier instruction or a lower lane of the same one, whichever
// cell of its row the node sits in.  (Adding cell k of every row with instruction k — the obvious layout — is wrong:
// a node is the first cell of one row and the second of another, and a lane would then see the additions of LATER
// rows made by an earlier instruction.  The emulator runs every LDS atomic as an instruction of its own for this
// reason.)  The same physical lane is therefore two things: "row lane" i evaluates row i, "pair lane" l adds the
// pairs l, 64 + l, 128 + l.  They meet in LDS words: the row lane publishes its current outcome as one ROW WORD (bit
// 4 + c: cell c is the first pick, bit 20 + c: the second), a pair lane reads the word of its pair's row, shifts it by
// its cell and masks it — its addend —, adds, and puts what the add returned into the pair's STAGING WORD stage[v];
// the row lane reads stage[3 i .. 3 i + 2]: the counter words as its row would see them if every earlier row of the
// tile had committed its current outcome.  (Pair lanes touch consecutive words, row lanes words at stride 3: no bank
// conflicts on either side.  The first version met in a 16-byte slot per row and was a third slower for it.)  If no lane's outcome changed, the words already hold the
// tile's commits (the additions stay); otherwise the additions are taken back (ds_sub_u32) and made again with the
// new outcomes.  Lane i is right once every lane below it is, so the loop ends after at most 65 evaluations, and
// the fixed point of an acyclic system is unique: it is the sequential result.  Emulator, bench-shaped scenarios
// (100k rows, 1000 brokers): 3.3 evaluations per tile on average — including the stretch of consecutive orphans
// that first fit put on one broker each, the chains the ticket forms need queues for: only near-ties ever change a
// pick, so a chain of 13 rows on one broker is right after two or three evaluations.
//
// Counter word of a node (uint32, LDS, 4 bytes x (n_max + 1)): count[n][0] in bits 0..15, count[n][1] in bits
// 16..31 (count[n][2] is never read for lists 3 wide: the last position has one candidate).  A pick is the minimum
// of three keys  count << 16 | visit position << 2 | cell:  for the first pick  x << 16 | tag  (one
// v_lshl_or_b32: the second count falls off the top), for the second  x & 0xffff0000 | tag  (one v_and_or_b32).
// Cells stay in the order the fill kernel left them; the visit order of KAS:188-200 / KAS:263-278 — ascending
// holders, rotated by abs(hash) mod set size — is in the tags: with rank = how many of the row's other holders are
// smaller, the first pick over three holders visits a holder at position (idx3 + rank) mod 3 and the second visits
// the two that are left in ascending (idx2 = 0) or descending order.  Tags are computed once per tile.  Rows per
// node stay below 65535 (KAS_RELAX_ROW_LIMIT): the 16-bit fields.
//
// Tiles whose 64 rows all hold three brokers in rows of the batch's width take the straight-line evaluation above;
// any other tile (the last tile of a topic, rows with fewer holders after a reduced replication factor, topics
// narrower than the batch) takes the same loop with per-lane list lengths.  The DUAL instance solves two usual tiles
// in a row as one tile of 128 rows (two rows per lane, six pair instructions): fewer LDS round trips per scenario,
// more LDS operations per row, 91 instead of 59 vector registers — for launches that do not fill the GPU
// (kas_relax_double_tiles).  HBM: mid rows in (8 B per row, read two tiles ahead), final rows out (12 B), broker ids
// from the L2-resident node table.
//
// Applicable (KasShape::relax_ok) to lists <= 3 wide with no topic hash of Integer.MIN_VALUE and fewer than 65535
// rows per node; a Context handed in (the CTX instances) must leave room for them in its columns 0 and 1, checked per
// scenario by the kernel.  Everything else keeps the ticket / round forms.
#pragma once

namespace kas {

#define KAS_RELAX_F0_ONE 0x1u          // count[n][0] += 1   (bits 0..15)
#define KAS_RELAX_F1_ONE 0x10000u      // count[n][1] += 1   (bits 16..31)
#define KAS_RELAX_F1_MASK 0xffff0000u
#define KAS_RELAX_PAD_WORD 0xfff0fff0u // counter word of the padding node: only ever gets + 0
#define KAS_RELAX_PICK_BITS 0x00010001u // (row word >> cell) & this: what the row adds to the counter word of that cell

// The row word: what a row's current outcome adds, for all of its cells at once — bit c set when cell c is the
// first pick, bit 16 + c when it is the second.  (word >> cell) & KAS_RELAX_PICK_BITS is the cell's addend.
KAS_DEV uint32_t relax_row_word(uint32_t w0, uint32_t w1) { return (KAS_RELAX_F0_ONE << w0) | (KAS_RELAX_F1_ONE << w1); }

#define KAS_RELAX_MAXP 12             // pair instructions of the largest tile (quad tiles: 256 rows x 3 cells / 64 lanes)
// A lane's pairs: pair v = 64 t + lane is cell v mod 3 of row v div 3 of the (double) tile; its staging word is
// stage[v] — the pair lanes touch consecutive words, the row lanes words 3 i .. 3 i + 2: both free of bank conflicts.
struct RelaxPairs {
  uint32_t* slot;                      // stage + lane: pair t's word is slot[64 t]
  const uint32_t* row[KAS_RELAX_MAXP];  // the row word of the pair's row
  uint32_t cell[KAS_RELAX_MAXP];        // the pair's cell
};

// per-topic constants of the picks (wave-uniform: scalar registers)
struct RelaxTopic {
  int32_t idx2, idx3;
  uint32_t vp3;                        // 4-bit field per rank: first pick over three holders, visit position << 2
  uint32_t ord2;                       // 4-bit field per rank: second pick's order among the two left, << 2
};

KAS_DEV RelaxTopic relax_topic(int32_t name_hash) {
  RelaxTopic t;
  t.idx2 = java_abs_mod(name_hash, 2);
  t.idx3 = java_abs_mod(name_hash, 3);
  t.vp3 = 0u; t.ord2 = 0u;
#pragma unroll
  for (int rank = 0; rank < 3; ++rank) {
    int32_t vp = t.idx3 + rank;                              // order[(idx + i) mod n] = sorted[i]  (KAS:193-197)
    vp -= vp >= 3 ? 3 : 0;
    t.vp3 |= (uint32_t)(vp << 2) << (4 * rank);
    // two holders left, ascending a < b: visited (a, b) when idx2 == 0, (b, a) when idx2 == 1
    t.ord2 |= (uint32_t)((t.idx2 ? 2 - rank : rank) << 2) << (4 * rank);
  }
  return t;
}

// The picks of one row, any list length (KAS:225-236 over KAS:263-278), cells in any order.  x[k] = counter word of
// cell k, valid[k] = the cell holds a broker, rank[k] = holders of the row below it.  Returns first pick | second
// pick << 2 as cell indices (0 where the list is shorter).
KAS_DEV int32_t relax_eval_generic(const uint32_t (&x)[3], const bool (&valid)[3], const int32_t (&rank)[3], int32_t Lp,
                                   const RelaxTopic& t) {
  const int32_t idx_m0 = Lp == 3 ? t.idx3 : (Lp == 2 ? t.idx2 : 0);
  uint32_t best = 0xffffffffu;
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    int32_t vp = idx_m0 + rank[k];
    vp -= vp >= Lp ? Lp : 0;
    const uint32_t key = ((x[k] & 0xffffu) << 8) | ((uint32_t)vp << 2) | (uint32_t)k;
    best = (valid[k] && key < best) ? key : best;
  }
  const int32_t w0 = Lp >= 1 ? (int32_t)(best & 3u) : 0;
  best = 0xffffffffu;
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    // two left (Lp == 3): ascending or descending by idx2; one left: no choice
    const int32_t ord = (Lp == 3 && t.idx2) ? 2 - rank[k] : rank[k];
    const uint32_t key = ((x[k] >> 16) << 8) | ((uint32_t)ord << 2) | (uint32_t)k;
    best = (valid[k] && k != w0 && key < best) ? key : best;
  }
  const int32_t w1 = Lp >= 2 ? (int32_t)(best & 3u) : 0;
  return w0 | (w1 << 2);
}

// One relaxation step of a tile, as the pair lanes see it: the rows' words are in rbuf; take the previous additions
// back, add the pairs in row-major order (NP instructions, lanes ascending), put what each add returned into the
// pair's staging word.  Afterwards every row lane finds in stage[3 i + c] the counter word of its cell c as its row
// would see it with the current outcomes of all earlier rows of the tile committed.
template <int NP>
KAS_DEV void relax_pairs(const RelaxPairs& pp, uint32_t* const (&padr)[KAS_RELAX_MAXP], uint32_t (&padd)[KAS_RELAX_MAXP], bool undo) {
  kasw::lockstep();                                          // the row words are written
  uint32_t nadd[NP];
#pragma unroll
  for (int t = 0; t < NP; ++t) nadd[t] = (*pp.row[t] >> pp.cell[t]) & KAS_RELAX_PICK_BITS;
  if (undo) {                                                // (wave-uniform)
#pragma unroll
    for (int t = 0; t < NP; ++t) kasw::lds_sub_u32(padr[t], padd[t]);
  }
  kasw::lockstep();
  uint32_t got[NP];
#pragma unroll
  for (int t = 0; t < NP; ++t) {
    got[t] = kasw::lds_add_rtn_u32(padr[t], nadd[t]);
    kasw::lockstep();                                        // (one instruction at a time, lanes in order: the hardware's order)
    padd[t] = nadd[t];
  }
#pragma unroll
  for (int t = 0; t < NP; ++t) pp.slot[64 * t] = got[t];
  kasw::lockstep();
}

// The six tags of a row whose three cells all hold a broker, from the order of its cells (tagtab, per topic).
struct RelaxTags { uint32_t t0[3], t1[3]; };

KAS_DEV RelaxTags relax_tags(const uint32_t (&c)[3], const uint32_t* tagtab) {
  const uint32_t oidx = ((c[0] - c[1]) >> 31) | (((c[0] - c[2]) >> 31) << 1) | (((c[1] - c[2]) >> 31) << 2);
  const uint32_t tw = tagtab[oidx];
  RelaxTags g;
#pragma unroll
  for (int q = 0; q < 3; ++q) { g.t0[q] = (tw >> (4 * q)) & 15u; g.t1[q] = (tw >> (12 + 4 * q)) & 15u; }
  return g;
}

// ... of a row whose cells are in ascending order (dword mid rows, KAS_FLAG_MID32): cell q has rank q — the topic's constants,
// wave-uniform (scalar registers), no table read and nothing computed per tile
KAS_DEV RelaxTags relax_tags_sorted(const RelaxTopic& t) {
  RelaxTags g;
#pragma unroll
  for (int q = 0; q < 3; ++q) {
    g.t0[q] = ((t.vp3 >> (4 * q)) & 0xcu) | (uint32_t)q;
    g.t1[q] = ((t.ord2 >> (4 * q)) & 0xcu) | (uint32_t)q;
  }
  return g;
}

#ifndef KAS_M32_SORTED_TAGS
#define KAS_M32_SORTED_TAGS(M32) (M32)
#endif
// The picks of a row with three holders: first pick | second pick << 2 (cell indices).
KAS_DEV int32_t relax_eval3(const uint32_t (&x)[3], const RelaxTags& g) {
  // first pick: count[.][0], first strictly smaller in visit order == minimum of (count, visit position)
  const uint32_t k00 = (x[0] << 16) | g.t0[0], k01 = (x[1] << 16) | g.t0[1], k02 = (x[2] << 16) | g.t0[2];
  const uint32_t kmin = k00 < k01 ? (k00 < k02 ? k00 : k02) : (k01 < k02 ? k01 : k02);
  const uint32_t w0 = kmin & 3u;
  // second pick: count[.][1] over the two that are left
  const uint32_t k10 = (x[0] & KAS_RELAX_F1_MASK) | g.t1[0], k11 = (x[1] & KAS_RELAX_F1_MASK) | g.t1[1],
                 k12 = (x[2] & KAS_RELAX_F1_MASK) | g.t1[2];
  const uint32_t lo = w0 == 0u ? k11 : k10, hi = w0 == 2u ? k11 : k12;
  const uint32_t w1 = (lo < hi ? lo : hi) & 3u;
  return (int32_t)(w0 | (w1 << 2));
}

// The final rows of a tile whose rows all hold three brokers, in two halves.  relax_list3: list position r takes cell
// w_r — bytes 2 w_r, 2 w_r + 1 of the mid row (v_perm_b32 selector, high half zero) — as node indices.  The broker ids
// of those nodes are ASKED FOR where the tile is decided (RelaxPend) and the rows go out one step later
// (relax_flush): the first version loaded the ids and stored the row on the spot, which made the wavefront wait for an L2
// round trip and then for the store's acknowledgement on every tile — a third of the kernel's time on its one chain.
// (cnt2 != nullptr: a Context goes back — count[.][2] of the last position's holder, never read by these lists, is kept
// as an increment per node beside the counter words)
template <class Raw>
KAS_DEV void relax_list3(const Raw& raw, int32_t oc, uint32_t (&l)[3], uint32_t* cnt2 = nullptr) {
  const uint32_t w0 = (uint32_t)oc & 3u, w1 = ((uint32_t)oc >> 2) & 3u, w2 = 3u - w0 - w1;
  l[0] = kasw::perm_bytes(raw.w[1], raw.w[0], 0x0c0c0100u + w0 * 0x0202u);
  l[1] = kasw::perm_bytes(raw.w[1], raw.w[0], 0x0c0c0100u + w1 * 0x0202u);
  l[2] = kasw::perm_bytes(raw.w[1], raw.w[0], 0x0c0c0100u + w2 * 0x0202u);
  if (cnt2) kasw::lds_add_u32(cnt2 + l[2], 1u);
}

// final rows on their way: the broker ids of NB rows per lane (rows p, p + 64), asked for (kasw::gload_u32_async) and
// not to be looked at before the step's kasw::wait_loads()
template <int NB>
struct RelaxPend {
  uint32_t id[NB][3];
  int32_t p;
  int32_t n;                           // rows per lane that are pending: 0, 1 or 2 (wave-uniform)
};

// A mid row asked for (row p must exist: the caller clamps) and, one step later, taken: the layout of mid_load_raw —
// FULLW, rows of the template width, as mid_width_of<W>() / 2 dwords, others cell by cell; cells past the row's width and
// rows past the topic's end read as KAS_MID_NONE.  Every word is asked for by exactly ONE unconditional statement (a cell
// the row does not have re-reads its last one and is masked when taken): a request under a condition would make the
// compiler merge two definitions of the variable with a register copy — of a register whose load is still in flight.
// (M32: dword mid rows, KAS_FLAG_MID32 — the row is ONE aligned dword whatever the topic's width)
template <int W, bool FULLW, bool M32 = false>
KAS_DEV void mid_request(MidRaw<W>& r, const uint16_t* mid, int32_t ow, int32_t p) {
  if constexpr (M32) {
    kasw::gload_u32_async<0>(r.w[0], mid, (uint32_t)p * 4u);
  } else if constexpr (FULLW) {                                    // (packed rows: the dword of a 6-byte row is 2-byte aligned)
    const uint32_t off = (uint32_t)p * (uint32_t)(2 * mid_width_of<W>());
    kasw::gload_u32_async<0>(r.w[0], mid, off);
    if constexpr (W == 3) kasw::gload_u16_async<4>(r.w[1], mid, off);
  } else {
    const uint32_t off = (uint32_t)p * (uint32_t)(2 * mid_width(ow));
#pragma unroll
    for (int k = 0; k < W; ++k) kasw::gload_u16_async<0>(r.w[k], mid, off + 2u * (uint32_t)(k < ow ? k : ow - 1));
  }
}
// (M32: the row's dword as it came, or the all-ones word — which reads as "no holder" three times — for a row that does not exist;
// mid_view takes it apart where the step uses it)
template <int W, bool FULLW, bool M32 = false>
KAS_DEV MidRaw<W> mid_take(const MidRaw<W>& r, int32_t ow, bool active) {
  MidRaw<W> o;
  if constexpr (M32) {
    o.w[0] = active ? r.w[0] : 0xffffffffu;
#pragma unroll
    for (int k = 1; k < W; ++k) o.w[k] = 0xffffffffu;
    return o;
  }
#pragma unroll
  for (int k = 0; k < W; ++k) o.w[k] = (active && (FULLW ? k < (W + 1) / 2 : k < ow)) ? r.w[k] : 0xffffffffu;
  return o;
}

// M32: a taken row in the layout of the 16-bit rows — FULLW: cells 0, 1 in w[0], cell 2 in w[1]; else a cell per word — its cells
// in ascending order.  "No holder" stays 0x7ff where FULLW (it is last in a sorted row, so a row holds three brokers iff its cell 2
// is below 0x7ff; the slow path turns it into KAS_MID_NONE) and is KAS_MID_NONE otherwise.
template <int W, bool FULLW, bool M32>
KAS_DEV MidRaw<W> mid_view(const MidRaw<W>& t, int32_t ow) {
  if constexpr (M32) {
    static_assert(W == 3, "dword mid rows: lists 3 wide");
    MidRaw<W> o;
    uint32_t f[3];
    mid32_fields(t.w[0], f);
    if constexpr (FULLW) {
      o.w[0] = f[0] | (f[1] << 16);
      o.w[1] = f[2] | 0xffff0000u;
      o.w[2] = 0xffffffffu;
    } else {
#pragma unroll
      for (int k = 0; k < 3; ++k) o.w[k] = (k < ow && f[k] != KAS_M32_NONE) ? f[k] : 0xffffffffu;
    }
    return o;
  } else {
    return t;
  }
}

// the pending rows go out (one 12-byte store each; C16, 16-bit cells: 6 bytes, in the place of the mid row they were made
// from); returns their digest
template <int NB, bool C16 = false>
KAS_DEV uint64_t relax_flush(RelaxPend<NB>& pend, int32_t* out, uint16_t* out16, uint32_t k) {
  uint64_t d = 0;
#pragma unroll
  for (int b = 0; b < NB; ++b) {
    if (b < pend.n) {                                        // (wave-uniform)
      RowW<3> o;
#pragma unroll
      for (int q = 0; q < 3; ++q) {
        o.v[q] = (int32_t)pend.id[b][q];
        d += kas_digest_cell(k, (uint32_t)(pend.p + 64 * b), (uint32_t)q, o.v[q]);
      }
#if !defined(KAS_TUNE_NO_ROW_STORES)                          // (tuning builds: the mid rows stay, the kernel can run again)
      if constexpr (C16) {
        uint16_t* row = out16 + (int64_t)(pend.p + 64 * b) * 3;
        store_u32_a2(row, (uint32_t)o.v[0] | ((uint32_t)o.v[1] << 16));
        row[2] = (uint16_t)o.v[2];
      } else {
        *reinterpret_cast<RowW<3>*>(out + (int64_t)(pend.p + 64 * b) * 3) = o;
      }
#endif
    }
  }
  pend.n = 0;
  return d;
}

// Sampled verification (KAS_PLAN_VERIFY_SAMPLE): a tile the relaxation has settled, evaluated AGAIN one row at a time —
// row i reads the counter words as rows 0 .. i - 1 left them, picks, adds its own counts with plain LDS adds — so that
// nothing depends on the order in which the LDS serves the lanes of one instruction.  Called with the tile's additions
// taken back; leaves the words as the sequential evaluation makes them (== what the relaxation had, if it was right).
// Returns the lanes whose row came out differently.  64 one-lane steps: ~3 us a tile.
// (the cells and their tags are made again from the mid row: nothing of the tile's evaluation has to stay in registers for
// this rarely taken path — with them passed in, the instance with tiles of 64 rows needed 66 instead of 60 registers)
KAS_DEV uint64_t relax_verify_rows(uint32_t* cnt, uint32_t w0_cells, uint32_t w1_cells, const uint32_t* tagtab, int32_t oc) {
  const int lane = kasw::lane();
  const uint32_t c[3] = {w0_cells & 0xffffu, w0_cells >> 16, w1_cells & 0xffffu};
  const RelaxTags g = relax_tags(c, tagtab);
  bool differs = false;
  for (int32_t i = 0; i < 64; ++i) {
    if (lane == i) {
      const uint32_t x[3] = {cnt[c[0]], cnt[c[1]], cnt[c[2]]};
      const int32_t os = relax_eval3(x, g);
      differs = os != oc;
      const uint32_t w0 = (uint32_t)os & 3u, w1 = ((uint32_t)os >> 2) & 3u;
      kasw::lds_add_u32(cnt + (w0 == 0u ? c[0] : (w0 == 1u ? c[1] : c[2])), KAS_RELAX_F0_ONE);
      kasw::lds_add_u32(cnt + (w1 == 0u ? c[0] : (w1 == 1u ? c[1] : c[2])), KAS_RELAX_F1_ONE);
    }
    kasw::lockstep();
  }
  return kasw::ballot(differs);
}

// DUAL: the instance that takes double tiles (a kernel of its own: it needs 91 vector registers, the instance without
// them 56 — two of its wavefronts fit where one wavefront of the fill kernel does)
// CTX: the instance for batches in which some scenario hands a Context in or wants it back (KAS:360-369): the counter
// words start from the Context's columns 0 and 1, a third array counts what the rows add to column 2, and all three go
// back at the end.  A scenario whose counters would leave the 16-bit fields is left to the round form (ord_flag).
// issue priority of the relaxation form's wavefront among the waves of its SIMD (s_setprio 0..3): its loop is ONE dependency
// chain, the fill kernel's wavefronts beside it have four tiles of independent work in flight each
#ifndef KAS_RELAX_PRIO
#define KAS_RELAX_PRIO 3
#endif
// ... and of a first-fit wavefront (kas_p4_kernel; wavefront 1 of kas_p4_order_kernel)
#ifndef KAS_P4_PRIO
#define KAS_P4_PRIO 2
#endif
// VERIFY: the instances for plans that ask for the sampled verification (KAS_PLAN_VERIFY_SAMPLE) — kernels of their own because
// the second evaluation keeps a tile's addresses and addends alive behind its loop: 67 instead of 60 vector registers for
// the instance with tiles of 64 rows, one register-file slot more than two of its wavefronts may take beside a fill wavefront
// C16: the instances for plans with 16-bit cells (KAS_FLAG_CELLS16): a final row is its node indices — no broker ids to ask
// for, nothing to wait for before the row goes out — stored over the mid row it was made from (round 6: VERIFY instances of
// these too — the second evaluation reads the tile's cells from registers, before the final rows take the mid rows' place)
// IDL (int32 cells; round 6): the scenario's broker ids are kept in the LDS (kas_relax_lds_ids: where 4 bytes a broker more fit
// comfortably) and a final row's ids are read from there — an LDS read (~100 cycles) on the kernel's one dependency chain where
// the L2-resident node table cost a round trip of ~700: order kernel alone 1.79 -> 1.46 ms per 1000 scenarios, twelve batches in
// flight 643k -> 726k scenarios/s (profiles/r06a_*).  IDL = false keeps the gather from the node table (asked for through
// kasw::gload_u32_async, waited for at the step's one s_waitcnt): broker counts whose ids do not fit.
// FS (kas_p4_order_kernel, round 6): first fit (P4) runs as a second wavefront of THIS workgroup (p4_scenario<W, 1, true>) and this
// wavefront follows it through fs[] — it asks for a tile's mid rows only when first fit is done with them (every row below the
// next window's first orphan is final), so that P5's chain starts while P4 is still placing orphans further down instead of
// behind a kernel boundary: a batch that has the GPU to itself lasts fill + max(P4, P5) instead of fill + P4 + P5.  A topic first
// fit fails (KAS:183-184) is abandoned where this wavefront stands: its rows' digest is dropped, fs[2] says that nothing more is
// written, and the first-fit wavefront pads the topic (nothing is returned for it: KAG:173-184).
// M32 (round 6): the instances for launches with dword mid rows (KAS_FLAG_MID32; kas_solver_body.h, mid32_pack): a row comes as
// ONE aligned dword — its holders in ascending order — instead of a dword and a halfword at 2-byte alignment, and because the
// cells are sorted a row's six tags are the topic's constants: no comparison of the cells, no tag table read.
// QUAD (round 6, with DUAL and M32): FOUR usual tiles in a row as one tile of 256 rows (four rows per lane, twelve pair
// instructions) — the same fixed point once more: fewer, longer steps for a launch whose latency is one scenario's chain.
template <int W, bool DUAL, bool CTX, bool VERIFY = false, bool C16 = false, bool IDL = false, bool FS = false, bool M32 = false, bool QUAD = false>
KAS_DEV void order_relax(const KasLaunch& a, int32_t s, unsigned char* lds_raw, uint64_t* fs = nullptr) {
  static_assert(W == 2 || W == 3, "counter words hold the counts of lists up to 3 wide");
  static_assert(!QUAD || (DUAL && M32), "quad tiles: an instance of the double-tile kind on dword mid rows");
  constexpr int NBT = QUAD ? 4 : (DUAL ? 2 : 1);             // tiles a step may take
  static_assert(!M32 || (W == 3 && !CTX && !VERIFY && !C16 && IDL), "dword mid rows: lists 3 wide, int32 cells with the ids in the LDS, no Context, no sampled verification");
  static_assert(!FS || (!CTX && !VERIFY), "first fit in the order kernel's workgroup: batches without a Context, no sampled verification");
  // (16-bit cells: with no broker ids to wait for the raised priority stops paying — 8 x 20 steps 815-834k scenarios/s at
  // priority 0 against 803-815k at 3, 8 x 40 steps 833-865k against 824-853k, same box, gpurun_out/r5pr2)
  if constexpr (KAS_RELAX_PRIO > 0 && !C16) kasw::set_priority<KAS_RELAX_PRIO>();
  const int lane = kasw::lane();
  const kas_scenario_desc sd = a.scen[s];
  const int32_t N = sd.n_nodes;
  const int32_t nmax = a.n_max > 0 ? a.n_max : 1;
  uint32_t* cnt = (uint32_t*)lds_raw;                       // [nmax + 1]: + the padding node's word
  uint32_t* tagtab = (uint32_t*)(lds_raw + kas_align16(4 * (int64_t)(nmax + 1)));   // [8] by the order of a row's three cells: its six tags
  uint32_t* rbuf = tagtab + 8;                              // [64 | 128] row words of the (double) tile
  uint32_t* stage = rbuf + 64 * NBT;                        // [192 | 384 | 768] by pair of the (double, quad) tile
  uint32_t* cnt2 = nullptr;                                 // (CTX) [nmax] what the rows add to count[n][2]
  uint32_t* idt = nullptr;                                  // (IDL, int32 cells) [nmax] the scenario's broker ids
  const int32_t* g_node_id = a.node_id + sd.node_off;
  const int64_t t_begin = kasw::clock_ticks();
  int32_t* g_ctx = nullptr;
  int32_t ccols = 0;
  if constexpr (CTX) {
    cnt2 = stage + 192 * NBT;
    if (sd.ctx_off >= 0 && sd.ctx_width > 0 && a.ctx != nullptr) {           // (wave-uniform)
      g_ctx = a.ctx + sd.ctx_off;
      ccols = sd.ctx_width < W ? sd.ctx_width : W;
      // columns 0 and 1 live in 16-bit fields: the Context's value + every row this scenario can add must fit
      const bool over = ctx_over_limit(g_ctx, N, sd.ctx_width, ccols < 2 ? ccols : 2, KAS_RELAX_ROW_LIMIT + 1u,
                                       ctx_gain_bound(a, sd), lane, 64);
      if (kasw::ballot(over) != 0ull) {
        if (lane == 0 && a.ord_flag) a.ord_flag[s] = 1;       // the round form takes this scenario
        return;
      }
    }
  }
  uint32_t seed0 = 0u, seed1 = 0u;                           // what the Context brings to the [0] / [1] fields
  for (int32_t n = lane; n < N; n += 64) {
    uint32_t w = 0u;
    if constexpr (CTX) {
      if (ccols > 0) w |= (uint32_t)g_ctx[(int64_t)n * sd.ctx_width];
      if (ccols > 1) w |= (uint32_t)g_ctx[(int64_t)n * sd.ctx_width + 1] << 16;
      cnt2[n] = 0u;
    }
    cnt[n] = w;
    seed0 += w & 0xffffu; seed1 += w >> 16;
  }
  // the broker ids of the final rows come from the LDS, not from the L2-resident node table: a final row then waits for an
  // LDS read (~100 cycles) where it waited for an L2 round trip (~700) on the kernel's one dependency chain
  constexpr bool LDSIDS = IDL && !C16;
  if constexpr (LDSIDS) {
    idt = stage + 192 * NBT + (CTX ? nmax : 0);
    for (int32_t n = lane; n < N; n += 64) idt[n] = (uint32_t)g_node_id[n];
  }
  if constexpr (CTX) {                                       // (wave-uniform from here on: scalar registers)
    seed0 = (uint32_t)kasw::uniform(kasw::wave_sum((int)seed0)); seed1 = (uint32_t)kasw::uniform(kasw::wave_sum((int)seed1));
  } else {
    seed0 = 0u; seed1 = 0u;
  }
  if (lane == 0) cnt[nmax] = KAS_RELAX_PAD_WORD;
  RelaxPairs pp;
  pp.slot = stage + lane;
#pragma unroll
  for (int t = 0; t < (QUAD ? 12 : 6); ++t) {
    const int32_t v = 64 * t + lane, r = v / 3;
    pp.row[t] = rbuf + r;
    pp.cell[t] = (uint32_t)(v - 3 * r);
  }
  uint32_t* const mine = stage + 3 * lane;                  // (row lane) my row's three words; + 192 for the second row
  kasw::lockstep();

  uint64_t digest = 0;
  int32_t n_tiles = 0, n_evals = 0, n_slow = 0, n_verified = 0;   // (wave-uniform)
  bool stuck = false;
  // Two checks on the form's one assumption (kas_wave.h, lds_add_rtn_u32).  Conservation, always on: every row with a first
  // (second) pick adds exactly one to some node's count[.][0] (count[.][1]) field, so when the last row has retired the
  // fields sum to what the Context brought plus the rows counted here — a lost or doubled addition cannot go unnoticed.
  // Sampled verification, on request (KAS_PLAN_VERIFY_SAMPLE(k): k tiles per topic): relax_verify_rows.
  uint32_t rows1 = 0u, rows2 = 0u;                           // rows that added to a [0] / [1] field (wave-uniform)
  bool unsound = false;
  const int32_t verify_k = VERIFY ? (int32_t)(a.flags >> 24) : 0;
  bool aborted = false;                                      // (FS) first fit failed a topic: the scenario ends there
  // (FS) wait until first fit is done with rows [0, upto) of topic k; false: the topic (or an earlier one) has failed
  auto rows_final = [&](int32_t k, int32_t upto) -> bool {
    if constexpr (FS) {
      int32_t idle = 0;
      for (;;) {
        kasw::repoll();
        const uint64_t f = kasw::load_shared_u64_lds(&fs[0]), fk = kasw::load_shared_u64_lds(&fs[1]);
        const int32_t ft = (int32_t)(f >> 32), fr = (int32_t)(uint32_t)f;
        if (kasw::ballot(fk <= (uint64_t)(uint32_t)k) != 0ull) return false;
        if (kasw::ballot(ft > k || (ft == k && fr >= upto)) != 0ull) return true;
        if (watchdog_poll(reinterpret_cast<uint32_t*>(&fs[3]), false, idle)) { stuck = true; return false; }
      }
    }
    return true;
  };
  for (int32_t k = 0; k < sd.topic_count; ++k) {
    const int32_t ti = sd.topic_begin + k;
    if constexpr (FS) {
      if (!rows_final(k, 0)) { aborted = true; break; }      // (first fit has reached this topic; an earlier failure ends the scenario)
    }
    if (a.topic_results[ti].status != KAS_OK) continue;     // (a failed or skipped topic emits nothing)
    const kas_topic_desc td = a.topics[ti];
    const int32_t P = td.n_partitions, ow = td.out_width;
    if (P <= 0) continue;
    uint64_t dtop = 0;                                       // the topic's digest: counted when the topic is through
    int32_t* out = C16 ? nullptr : a.out + td.out_off;
    uint16_t* out16 = C16 ? reinterpret_cast<uint16_t*>(a.out) + td.out_off : nullptr;
    const uint16_t* mid = C16 ? out16 : (M32 ? reinterpret_cast<const uint16_t*>(out + (int64_t)P * ow - P) : mid_base(out, P, ow));
    const RelaxTopic rt = relax_topic(td.name_hash);
    const RelaxTags gsorted = relax_tags_sorted(rt);        // (M32)
    constexpr uint32_t NONE_FROM = M32 ? KAS_M32_NONE : 0x8000u;   // a cell at or above this holds no broker (cells of a row that exists)
    const int32_t nt = (P + 63) >> 6;
    // the topic's tags by the order of a row's cells: bit 0 = cell 0 < cell 1, bit 1 = cell 0 < cell 2, bit 2 = cell 1 <
    // cell 2;  word = first-pick tags of cells 0..2 (4 bits each), then the second-pick tags
    kasw::lockstep();
    if (lane < 8) {
      const int32_t ab = lane & 1, ac = (lane >> 1) & 1, bc = (lane >> 2) & 1;
      const int32_t rank[3] = {(1 - ab) + (1 - ac), ab + (1 - bc), ac + bc};
      uint32_t w = 0u;
#pragma unroll
      for (int q = 0; q < 3; ++q) {
        w |= (((rt.vp3 >> (4 * rank[q])) & 0xcu) | (uint32_t)q) << (4 * q);
        w |= (((rt.ord2 >> (4 * rank[q])) & 0xcu) | (uint32_t)q) << (12 + 4 * q);
      }
      tagtab[lane] = w;
    }
    kasw::lockstep();
    // (rows of the batch's width, the usual case, and narrower ones: two instances of the topic's row loop)
    const uint16_t* umid = kasw::uniform_ptr(mid);
    const int32_t* uid = kasw::uniform_ptr(g_node_id);
    auto topic_rows = [&](auto fullw_tag) {
      constexpr bool FULLW = decltype(fullw_tag)::value;
      // Mid rows: raw[] = the rows of this step (one tile; two for the instance with double tiles), in registers; nx[] =
      // the rows behind them, asked for one step ago.  Every global access of a step sits at ONE point of it, its end:
      // there the wavefront waits (the only place it does) for what it asked for a whole step ago — the next rows, the
      // broker ids of the step's predecessor, whose final rows then go out — and makes its new requests; nothing before
      // the same point of the next step looks at them.  The requests are loads the compiler does not see
      // (kasw::gload_*_async: its own wait insertion put a vmcnt(0) behind the requests of the same iteration).
      constexpr int NB = NBT;
      const int32_t vstride = verify_k > 0 ? (nt / verify_k > 0 ? nt / verify_k : 1) : 0;   // every vstride-th tile is verified,
      const int32_t voff = vstride > 0 ? s % vstride : 0;                                    // starting at a tile of the scenario's own
      auto row_exists = [&](int32_t t) -> bool { return ((t << 6) + lane) < P; };
      auto request_tile = [&](MidRaw<W>& r, int32_t t) {
        const int32_t pn = (t << 6) + lane;
        mid_request<W, FULLW, M32>(r, umid, ow, pn < P ? pn : 0);
      };
      MidRaw<W> raw[NB], nx[NB];
      RelaxPend<NB> pend;
      pend.n = 0; pend.p = 0;
#pragma unroll
      for (int b = 0; b < NB; ++b) {
#pragma unroll
        for (int q = 0; q < W; ++q) { raw[b].w[q] = 0xffffffffu; nx[b].w[q] = 0xffffffffu; }
#pragma unroll
        for (int q = 0; q < 3; ++q) pend.id[b][q] = 0u;
      }
#pragma unroll
      for (int b = 0; b < NB; ++b) request_tile(nx[b], b);
      kasw::wait_loads();
#pragma unroll
      for (int b = 0; b < NB; ++b) {
#pragma unroll
        for (int q = 0; q < (M32 ? 1 : W); ++q) kasw::arrived(nx[b].w[q]);
        raw[b] = mid_take<W, FULLW, M32>(nx[b], ow, row_exists(b));
      }
#pragma unroll
      for (int b = 0; b < NB; ++b) request_tile(nx[b], NB + b);
      for (int32_t tile = 0; tile < nt;) {
        const int32_t p = (tile << 6) + lane;
        const bool active = p < P;
        // ---- cells of my row (row lane): node index or KAS_MID_NONE (0xffff, bit 15)
        // (M32: the rows' dwords taken apart here, into the layout of the 16-bit rows — mid_view)
        const MidRaw<W> ra = mid_view<W, FULLW, M32>(raw[0], ow);
        uint32_t c[3];
        if constexpr (FULLW) {
          c[0] = ra.w[0] & 0xffffu; c[1] = ra.w[0] >> 16; c[2] = W == 3 ? (ra.w[1] & 0xffffu) : KAS_MID_NONE;
        } else {
#pragma unroll
          for (int q = 0; q < 3; ++q) c[q] = q < W ? (ra.w[q] & 0xffffu) : KAS_MID_NONE;
        }
        // the usual tile: 64 rows, three holders each, rows of the batch's width
        bool fast = false;
        uint32_t* padr[KAS_RELAX_MAXP];
        uint32_t padd[KAS_RELAX_MAXP] = {0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u};   // what my pairs added last
        int32_t step = 1;                                      // tiles this step takes (wave-uniform)
        int32_t req_n = 0;                                     // rows per lane whose final rows this step asks the ids for
        uint32_t req_l[NB][3];                                 // ... their lists as node indices
#pragma unroll
        for (int b = 0; b < NB; ++b)
#pragma unroll
          for (int q = 0; q < 3; ++q) req_l[b][q] = 0u;
        if constexpr (W == 3) {
          if (FULLW && ((tile + 1) << 6) <= P)
            fast = M32 ? kasw::ballot(c[2] >= NONE_FROM) == 0ull      // (sorted: a row holds three brokers iff its last cell does)
                       : kasw::ballot(((c[0] | c[1] | c[2]) & 0x8000u) != 0u) == 0ull;
          // ---- four usual tiles in a row: one quad tile of 256 rows, lane i evaluates rows i, 64 + i, 128 + i, 192 + i (twelve
          // pair instructions, row-major); anything else falls through to a single tile
          if constexpr (QUAD) if (fast && ((tile + 4) << 6) <= P) {
            MidRaw<W> rv[4];
            uint32_t cc[4][3];
            rv[0] = ra;
#pragma unroll
            for (int q = 0; q < 3; ++q) cc[0][q] = c[q];
            bool short_row = false;
#pragma unroll
            for (int b = 1; b < 4; ++b) {
              rv[b] = mid_view<W, FULLW, M32>(raw[b], ow);
              cc[b][0] = rv[b].w[0] & 0xffffu; cc[b][1] = rv[b].w[0] >> 16; cc[b][2] = rv[b].w[1] & 0xffffu;
              short_row = short_row || cc[b][2] >= NONE_FROM;
            }
            if (kasw::ballot(short_row) == 0ull) {
              n_tiles += 4;
              kasw::lockstep();                                // (the previous tile's words have been read)
#pragma unroll
              for (int b = 0; b < 4; ++b)
#pragma unroll
                for (int q = 0; q < 3; ++q) mine[192 * b + q] = cc[b][q];
              kasw::lockstep();
#pragma unroll
              for (int t = 0; t < 12; ++t) padr[t] = cnt + pp.slot[64 * t];
              uint32_t xq[4][3];
#pragma unroll
              for (int b = 0; b < 4; ++b)
#pragma unroll
                for (int q = 0; q < 3; ++q) xq[b][q] = cnt[cc[b][q]];
              kasw::lockstep();
              int32_t pq[4] = {-1, -1, -1, -1};
              for (int32_t it = 0;; ++it) {
                n_evals += 4;
                if (it > 258) { stuck = true; break; }         // (row i is right after evaluation i + 1: 257 suffice)
                int32_t oq[4];
                bool moved = false;
#pragma unroll
                for (int b = 0; b < 4; ++b) { oq[b] = relax_eval3(xq[b], gsorted); moved = moved || oq[b] != pq[b]; }
                if (kasw::ballot(moved) == 0ull) break;
#pragma unroll
                for (int b = 0; b < 4; ++b) rbuf[64 * b + lane] = relax_row_word((uint32_t)oq[b] & 3u, (uint32_t)oq[b] >> 2);
                relax_pairs<12>(pp, padr, padd, it > 0);
#pragma unroll
                for (int b = 0; b < 4; ++b) {
#pragma unroll
                  for (int q = 0; q < 3; ++q) xq[b][q] = mine[192 * b + q];
                  pq[b] = oq[b];
                }
              }
              rows1 += 256u; rows2 += 256u;
#pragma unroll
              for (int b = 0; b < 4; ++b) relax_list3(rv[b], pq[b] < 0 ? 4 : pq[b], req_l[b], cnt2);
              req_n = 4;
              step = 4;
            }
          }
          // ---- two usual tiles in a row: one double tile of 128 rows, lane i evaluates rows i and 64 + i.  The same
          // fixed point (row-major pairs over six instructions), twice the work per LDS round trip.
          if constexpr (DUAL && !QUAD) if (fast && ((tile + 2) << 6) <= P) {
            const MidRaw<W> rb = mid_view<W, FULLW, M32>(raw[NB - 1], ow);
            const uint32_t cb[3] = {rb.w[0] & 0xffffu, rb.w[0] >> 16, rb.w[1] & 0xffffu};
            if ((M32 ? kasw::ballot(cb[2] >= NONE_FROM) : kasw::ballot(((cb[0] | cb[1] | cb[2]) & 0x8000u) != 0u)) == 0ull) {
              n_tiles += 2;
              kasw::lockstep();                                // (the previous tile's words have been read)
#pragma unroll
              for (int q = 0; q < 3; ++q) { mine[q] = c[q]; mine[192 + q] = cb[q]; }
              kasw::lockstep();
#pragma unroll
              for (int t = 0; t < 6; ++t) padr[t] = cnt + pp.slot[64 * t];
              uint32_t xa[3] = {cnt[c[0]], cnt[c[1]], cnt[c[2]]};
              uint32_t xb[3] = {cnt[cb[0]], cnt[cb[1]], cnt[cb[2]]};
              const RelaxTags ga = KAS_M32_SORTED_TAGS(M32) ? gsorted : relax_tags(c, tagtab), gb = KAS_M32_SORTED_TAGS(M32) ? gsorted : relax_tags(cb, tagtab);
              kasw::lockstep();
              int32_t pa = -1, pb = -1;
              for (int32_t it = 0;; ++it) {
                n_evals += 2;
                if (it > 130) { stuck = true; break; }         // (row i is right after evaluation i + 1: 129 suffice)
                const int32_t oa = relax_eval3(xa, ga), ob = relax_eval3(xb, gb);
                if (kasw::ballot(oa != pa || ob != pb) == 0ull) break;
                rbuf[lane] = relax_row_word((uint32_t)oa & 3u, (uint32_t)oa >> 2);
                rbuf[64 + lane] = relax_row_word((uint32_t)ob & 3u, (uint32_t)ob >> 2);
                relax_pairs<6>(pp, padr, padd, it > 0);
#pragma unroll
                for (int q = 0; q < 3; ++q) { xa[q] = mine[q]; xb[q] = mine[192 + q]; }
                pa = oa; pb = ob;
              }
              rows1 += 128u; rows2 += 128u;
              if (VERIFY && vstride > 0 && ((tile % vstride) == voff || ((tile + 1) % vstride) == voff)) {   // (wave-uniform)
                n_verified += 2;
#pragma unroll
                for (int t = 0; t < 6; ++t) kasw::lds_sub_u32(padr[t], padd[t]);
                kasw::lockstep();
                if (relax_verify_rows(cnt, ra.w[0], ra.w[1], tagtab, pa) != 0ull) unsound = true;
                if (relax_verify_rows(cnt, rb.w[0], rb.w[1], tagtab, pb) != 0ull) unsound = true;
              }
              relax_list3(ra, pa < 0 ? 4 : pa, req_l[0], cnt2);
              relax_list3(rb, pb < 0 ? 4 : pb, req_l[NB - 1], cnt2);
              req_n = 2;
              step = 2;
            }
          }
        }
        if (step == 1) {
          n_tiles += 1;
          // ---- hand the cells to the pair lanes
          kasw::lockstep();                                    // (the previous tile's words have been read)
#pragma unroll
          for (int q = 0; q < 3; ++q) mine[q] = c[q];
          kasw::lockstep();
          bool usual = false;
          if constexpr (W == 3) usual = fast;
          if (usual) {
            if constexpr (W == 3) {
#pragma unroll
              for (int t = 0; t < 3; ++t) padr[t] = cnt + pp.slot[64 * t];
              // counter words of my cells as the previous tile left them
              uint32_t x[3] = {cnt[c[0]], cnt[c[1]], cnt[c[2]]};
              const RelaxTags g = KAS_M32_SORTED_TAGS(M32) ? gsorted : relax_tags(c, tagtab);   // my row's six tags from the order of its cells
              kasw::lockstep();
              int32_t oc_prev = -1;
              for (int32_t it = 0;; ++it) {
                n_evals += 1;
                // (lane i is right after evaluation i + 1, so 65 evaluations always suffice: more means the LDS did not
                // hand the additions out in lane order — give up with a status instead of looping)
                if (it > 66) { stuck = true; break; }
                const int32_t oc = relax_eval3(x, g);
                if (kasw::ballot(oc != oc_prev) == 0ull) break;  // nobody's outcome moved: the words hold the tile's commits
                rbuf[lane] = relax_row_word((uint32_t)oc & 3u, (uint32_t)oc >> 2);
                relax_pairs<3>(pp, padr, padd, it > 0);
                x[0] = mine[0]; x[1] = mine[1]; x[2] = mine[2];
                oc_prev = oc;
              }
              rows1 += 64u; rows2 += 64u;
              if (VERIFY && vstride > 0 && (tile % vstride) == voff) {     // (wave-uniform)
                n_verified += 1;
#pragma unroll
                for (int t = 0; t < 3; ++t) kasw::lds_sub_u32(padr[t], padd[t]);
                kasw::lockstep();
                if (relax_verify_rows(cnt, ra.w[0], ra.w[1], tagtab, oc_prev) != 0ull) unsound = true;
              }
              // ---- the final row: its list as node indices now, broker ids and the store one step later
              relax_list3(ra, oc_prev < 0 ? 4 : oc_prev, req_l[0], cnt2);
              req_n = 1;
            }
          } else {
            // ---- any other tile: per-lane list lengths
            n_slow += 1;
            if constexpr (M32) {                              // ("no holder" as the code below knows it)
#pragma unroll
              for (int q = 0; q < 3; ++q) c[q] = c[q] >= NONE_FROM ? KAS_MID_NONE : c[q];
            }
#pragma unroll
            for (int t = 0; t < 3; ++t) {
              const uint32_t n = pp.slot[64 * t];
              padr[t] = cnt + (n < (uint32_t)nmax ? n : (uint32_t)nmax);   // no holder: the padding node, and + 0
            }
            uint32_t x[3];
#pragma unroll
            for (int q = 0; q < 3; ++q) x[q] = cnt[c[q] < (uint32_t)nmax ? c[q] : (uint32_t)nmax];
            kasw::lockstep();
            bool valid[3];
            int32_t rank[3], Lp = 0;
#pragma unroll
            for (int q = 0; q < 3; ++q) { valid[q] = active && (c[q] & 0x8000u) == 0u; Lp += valid[q] ? 1 : 0; }
#pragma unroll
            for (int q = 0; q < 3; ++q) {
              rank[q] = 0;
#pragma unroll
              for (int j = 0; j < 3; ++j) rank[q] += (j != q && valid[j] && c[j] < c[q]) ? 1 : 0;
            }
            rows1 += (uint32_t)kasw::popc(kasw::ballot(Lp >= 1)); rows2 += (uint32_t)kasw::popc(kasw::ballot(Lp >= 2));
            int32_t oc_prev = -1;
            for (int32_t it = 0;; ++it) {
              n_evals += 1;
              if (it > 66) { stuck = true; break; }
              const int32_t oc = relax_eval_generic(x, valid, rank, Lp, rt);
              if (kasw::ballot(oc != oc_prev) == 0ull) break;
              const uint32_t w0 = (uint32_t)oc & 3u, w1 = ((uint32_t)oc >> 2) & 3u;
              rbuf[lane] = (Lp >= 1 ? KAS_RELAX_F0_ONE << w0 : 0u) | (Lp >= 2 ? KAS_RELAX_F1_ONE << w1 : 0u);
              relax_pairs<3>(pp, padr, padd, it > 0);
#pragma unroll
              for (int q = 0; q < 3; ++q) x[q] = mine[q];
              oc_prev = oc;
            }
            if (active && oc_prev >= 0) {
              const int32_t w0 = oc_prev & 3, w1 = (oc_prev >> 2) & 3, w2 = 3 - w0 - w1;
              const int32_t w[3] = {w0, w1, w2};
#pragma unroll
              for (int r = 0; r < W; ++r) {
                if (r < ow) {
                  const uint32_t cell = w[r] == 0 ? c[0] : (w[r] == 1 ? c[1] : c[2]);
                  const int32_t id = r < Lp ? (C16 ? (int32_t)cell : (LDSIDS ? (int32_t)idt[cell] : g_node_id[cell])) : -1;
                  if constexpr (CTX) { if (r == 2 && r < Lp) kasw::lds_add_u32(cnt2 + cell, 1u); }
                  if constexpr (C16) out16[(int64_t)p * ow + r] = (uint16_t)id;
                  else out[(int64_t)p * ow + r] = id;
                  if (r < Lp) dtop += kas_digest_cell((uint32_t)k, (uint32_t)p, (uint32_t)r, id);
                }
              }
            }
          }
        }
        // ---- the step's global accesses: wait for what was asked for a step ago ...
        tile += step;
        kasw::wait_loads();
#pragma unroll
        for (int b = 0; b < NB; ++b) {
#pragma unroll
          for (int q = 0; q < (M32 ? 1 : W); ++q) kasw::arrived(nx[b].w[q]);
#pragma unroll
          for (int q = 0; q < 3; ++q) kasw::arrived(pend.id[b][q]);
        }
        // ... the next rows move up,
        if constexpr (NB == 1) {
          raw[0] = mid_take<W, FULLW, M32>(nx[0], ow, row_exists(tile));
        } else if constexpr (NB == 4) {
          if (step == 4) {
#pragma unroll
            for (int b = 0; b < 4; ++b) raw[b] = mid_take<W, FULLW, M32>(nx[b], ow, row_exists(tile + b));
          } else {
            raw[0] = raw[1]; raw[1] = raw[2]; raw[2] = raw[3]; raw[3] = mid_take<W, FULLW, M32>(nx[0], ow, row_exists(tile + 3));
          }
        } else {
          if (step == 2) {
            raw[0] = mid_take<W, FULLW, M32>(nx[0], ow, row_exists(tile)); raw[1] = mid_take<W, FULLW, M32>(nx[1], ow, row_exists(tile + 1));
          } else {
            raw[0] = raw[1]; raw[1] = mid_take<W, FULLW, M32>(nx[0], ow, row_exists(tile + 1));
          }
        }
        // the previous step's final rows go out,
        dtop += relax_flush<NB, C16>(pend, out, out16, (uint32_t)k);
        // and the requests are made: the broker ids of this step's final rows, the mid rows two steps on
        // (every in-flight register has ONE requesting statement, executed on every path: a step without final rows of
        // its own asks for node 0's id, a single-tile step of the double-tile instance asks for a tile it had already)
#pragma unroll
        for (int b = 0; b < NB; ++b)
#pragma unroll
          for (int q = 0; q < 3; ++q) {
            if constexpr (C16) pend.id[b][q] = req_l[b][q];           // (the cell IS the node index)
            else if constexpr (LDSIDS) pend.id[b][q] = idt[req_l[b][q]];
            else kasw::gload_u32_async<0>(pend.id[b][q], uid, req_l[b][q] << 2);
          }
        pend.p = p; pend.n = req_n;
        // The instance with tiles of 64 rows — launches that fill the GPU with wavefronts — sends the final rows out in the
        // step that decided them: it waits for the broker ids here, where the other wavefronts of the SIMD have work to
        // issue.  Measured (experiments/README.md, round 5): with the GPU full the deferred rows cost 5 % (2000 x 4 in
        // flight 588k -> 620k scenarios/s, 4000 x 3 590k -> 630k, 1000 x 8 605k -> 614k), while a batch alone gains 15 %
        // from them (order kernel 1.79 -> 1.51 ms) — and a batch alone is what the double-tile instance is launched for.
        if constexpr (!DUAL) {
          if constexpr (!C16 && !LDSIDS) {
            kasw::wait_loads();
#pragma unroll
            for (int b = 0; b < NB; ++b)
#pragma unroll
              for (int q = 0; q < 3; ++q) kasw::arrived(pend.id[b][q]);
          }
          dtop += relax_flush<NB, C16>(pend, out, out16, (uint32_t)k);
        }
        // (FS: the rows about to be asked for must be final — first fit is usually far ahead, and this is two LDS reads)
        if constexpr (FS) {
          // (a topic that has failed is left through the loop's own exit: a second way out of the loop BEHIND the requests below
          // would make the compiler copy registers whose loads are in flight — tools/check_async_loads.py)
          const int32_t upto = (tile + 2 * NB) << 6;
          if (tile < nt && !rows_final(k, upto < P ? upto : P)) { aborted = true; tile = nt; }
        }
#pragma unroll
        for (int b = 0; b < NB; ++b) request_tile(nx[b], tile + NB + b);
      }
      // the topic's last rows (and no request is left outstanding: its register would be written behind our back)
      kasw::wait_loads();
#pragma unroll
      for (int b = 0; b < NB; ++b) {
#pragma unroll
        for (int q = 0; q < (M32 ? 1 : W); ++q) kasw::arrived(nx[b].w[q]);
#pragma unroll
        for (int q = 0; q < 3; ++q) kasw::arrived(pend.id[b][q]);
      }
      if (aborted) pend.n = 0;                               // (FS: the topic has failed, nothing more is written)
      dtop += relax_flush<NB, C16>(pend, out, out16, (uint32_t)k);
    };
    if constexpr (FS) {
      // the first tiles' rows must be final before they are asked for
      const int32_t upto0 = (2 * NBT) << 6;
      if (!rows_final(k, upto0 < P ? upto0 : P)) { aborted = true; break; }
    }
    if (ow == W) topic_rows(std::true_type{});
    else topic_rows(std::false_type{});
    if (aborted) break;
    digest += dtop;
  }
  if constexpr (FS) {
    if (aborted) {
      // nothing of mine is on its way to the failed topic's rows any more: say so, the first-fit wavefront pads them
      kasw::wave_sync();
      if (lane == 0) kasw::store_shared_u64_lds(&fs[2], 1ull);
    }
    // the records of this scenario are first fit's until it has passed the last topic; the digest goes in behind them
    int32_t idle = 0;
    for (;;) {
      kasw::repoll();
      if (kasw::ballot((int32_t)(kasw::load_shared_u64_lds(&fs[0]) >> 32) >= sd.topic_count) != 0ull) break;
      if (watchdog_poll(reinterpret_cast<uint32_t*>(&fs[3]), false, idle)) { stuck = true; break; }
    }
  }
  {
    // conservation: the fields of every node's word against the rows that added to them (and the padding node's word untouched)
    kasw::lockstep();
    uint32_t f0 = 0u, f1 = 0u;
    for (int32_t n = lane; n < N; n += 64) { const uint32_t w = cnt[n]; f0 += w & 0xffffu; f1 += w >> 16; }
    const uint32_t d0 = (uint32_t)kasw::wave_sum((int)f0) - seed0, d1 = (uint32_t)kasw::wave_sum((int)f1) - seed1;
    if (!stuck && !aborted && (d0 != rows1 || d1 != rows2 || kasw::ballot(cnt[nmax] != KAS_RELAX_PAD_WORD) != 0ull)) unsound = true;
  }
  if constexpr (CTX) {
    // the Context goes back (KAS:360-369): every row has retired, the words hold every commit
    kasw::lockstep();
    for (int32_t n = lane; n < N && ccols > 0; n += 64) {
      const uint32_t w = cnt[n];
      int32_t* row = g_ctx + (int64_t)n * sd.ctx_width;
      row[0] = (int32_t)(w & 0xffffu);
      if (ccols > 1) row[1] = (int32_t)(w >> 16);
      if (ccols > 2) row[2] += (int32_t)cnt2[n];
    }
  }
  const uint64_t dsum = kasw::wave_sum_u64(digest);
  if (lane == 0) {
    a.scenario_results[s].digest = dsum;
    if (stuck || unsound) {
      a.scenario_results[s].status = KAS_FAIL_WATCHDOG;
      a.scenario_results[s].fail_topic = -1; a.scenario_results[s].fail_partition = -1;
    }
    if (a.stats) {
      int64_t* st = a.stats + (int64_t)s * KAS_STATS_PER_SCENARIO;
      st[8] = kasw::clock_ticks() - t_begin; st[9] = n_evals; st[12] = n_tiles; st[13] = n_slow; st[10] = n_verified; st[11] = unsound ? 1 : 0;
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// kas_p4_order_kernel, one scenario: first fit (P4) and the relaxation form of P5 in ONE workgroup of two wavefronts —
// wavefront 0 is order_relax<.., FS>, wavefront 1 is p4_scenario<W, 1, FS> on the hand-over of the fill kernel
// (KAS_FLAG_SPLIT_P4: loads after the sticky fill, the chunks' orphan lists), and fs[] (four 8-byte words of LDS between
// the two carve-ups) is how the first follows the second.  LDS: kas_p4_order_lds.
// ---------------------------------------------------------------------------------------------------------------------
template <int W, bool DUAL, bool C16, bool IDL, bool M32 = false, bool QUAD = false>
KAS_DEV void p4_order_scenario(const KasLaunch& a, int32_t s, unsigned char* lds_raw) {
  const int32_t nmax = a.n_max > 0 ? a.n_max : 1;
  const int32_t off_fs = kas_align16(kas_order_relax_lds(nmax, QUAD ? 2 : (DUAL ? 1 : 0), 0, (IDL && !C16) ? 1 : 0));
  uint64_t* fs = reinterpret_cast<uint64_t*>(lds_raw + off_fs);
  if (kasw::tid() < 4) fs[kasw::tid()] = kasw::tid() == 1 ? 0x7fffffffull : 0ull;   // rows final: none; failed topic: none; no answer; no watchdog
  kasw::sync();                                              // (the one workgroup barrier: both wavefronts pass it exactly once)
  if (kasw::wave_id() == 0) {
    order_relax<W, DUAL, false, false, C16, IDL, true, M32, QUAD>(a, s, lds_raw, fs);
  } else {
    if constexpr (KAS_P4_PRIO > 0) kasw::set_priority<KAS_P4_PRIO>();
    p4_scenario<W, 1, true, M32 ? 1 : 0>(a, s, lds_raw + off_fs + 32, fs);
  }
}

}  // namespace kas
