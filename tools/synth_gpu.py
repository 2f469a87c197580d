"""Synthetic genome-scale workload generated ON the GPU with torch (bench / at-scale test tooling).

hg38 is not on the GPU box, so BASELINE.json's config 2 ("10 M x 100 bp SE, -M C:T, hg38") runs on
a stand-in: 24 contigs with hg38's chromosome sizes (3.09 Gbp), uniform ACGT, N gaps
(one "centromere" per contig plus short gaps) and a planted diverged repeat family.  Because the
seed index hashes 3-letter-reduced 16-mers every 4th base of both strands, a uniform 3.1 Gbp
genome already gives hg38's mean index density (~36 locations per seed).
Reads: uniform start in non-N sequence, 50 % reverse strand, revcomp THEN convert (directional
protocol), per-base substitution rate.  Everything is seeded; torch only provides device memory
and elementwise ops -- none of this is part of the product path.
"""
import numpy as np
import torch

HG38_SIZES = [248956422, 242193529, 198295559, 190214555, 181538259, 170805979, 159345973, 145138636,
              138394717, 133797422, 135086622, 133275309, 114364328, 107043718, 101991189, 90338345,
              83257441, 80373285, 58617616, 64444167, 46709983, 50818468, 156040895, 57227415]
NAMES = ["chr%d" % i for i in range(1, 23)] + ["chrX", "chrY"]
MARGIN = 400


def _codes(params):
    """base id (A,C,G,T = 0..3) -> 2-bit code, forward and for the complement."""
    al = [params.c.alphabet[ord(b)] for b in "ACGT"]
    rv = [params.c.rev_alphabet[ord(b)] for b in "ACGT"]
    return al, rv


def _pack(codes_u8):
    """uint8 codes, length multiple of 32 -> int64 words, base 0 in the top two bits."""
    sh = torch.arange(31, -1, -1, device=codes_u8.device, dtype=torch.int64) * 2
    return (codes_u8.view(-1, 32).to(torch.int64) << sh).sum(dim=1)


class Genome:
    pass


# A repeat landscape shaped like the human genome's (shares of hg38 from RepeatMasker-style summaries, rounded): interspersed families as
# (name, consensus length, share of the genome, divergence range of a copy from the consensus, copy length classes [(fraction of copies,
# fraction of the consensus kept -- the 3' end, as truncated L1 copies keep it)]), plus tandem satellites.  About 45 % of the genome.
REALISTIC_FAMILIES = [
    ("Alu", 300, 0.105, (0.05, 0.18), [(1.0, 1.0)]),
    ("L1", 6000, 0.17, (0.03, 0.25), [(0.55, 0.08), (0.25, 0.15), (0.12, 0.3), (0.06, 0.6), (0.02, 1.0)]),
    ("MIR", 260, 0.03, (0.2, 0.32), [(1.0, 0.7)]),
    ("L2", 3300, 0.035, (0.2, 0.32), [(0.7, 0.1), (0.3, 0.25)]),
    ("LTR", 1000, 0.085, (0.08, 0.25), [(0.6, 0.5), (0.4, 1.0)]),
    ("DNA", 800, 0.03, (0.15, 0.3), [(0.7, 0.3), (0.3, 0.8)]),
]
REALISTIC_SATELLITE = (171, 0.03, 0.02, 200_000)  # monomer length, share of the genome, divergence between monomers, array length


def _plant_realistic(ids, L, scale_share, g, rng, device, consensus):
    """Overwrites stretches of one contig (ids, uint8 base ids) with diverged copies of the families' consensus sequences."""
    for (name, clen, share, (d_lo, d_hi), classes), cons in zip(REALISTIC_FAMILIES, consensus):
        for frac, keep in classes:
            ln = max(32, int(clen * keep))
            n = int(L * share * scale_share * frac / ln)
            if n <= 0 or L <= 2 * ln:
                continue
            starts = torch.from_numpy(rng.integers(0, L - ln, size=n)).to(device)
            unit = cons[clen - ln:].repeat(n)  # the 3' part of the consensus
            div = d_lo + (d_hi - d_lo) * torch.rand(n, generator=g, device=device)
            mut = torch.rand(n * ln, generator=g, device=device) < div.repeat_interleave(ln)
            unit = torch.where(mut, torch.randint(0, 4, unit.shape, generator=g, device=device, dtype=torch.uint8), unit)
            rc = torch.rand(n, generator=g, device=device) < 0.5  # either orientation
            unit = unit.view(n, ln)
            unit = torch.where(rc[:, None], (3 - unit).flip(1), unit).reshape(-1)
            idx = (starts[:, None] + torch.arange(ln, device=device)[None, :]).reshape(-1)
            ids[idx] = unit
            del starts, unit, mut, idx
    mono, share, sdiv, alen = REALISTIC_SATELLITE
    n_arr = max(1, int(L * share * scale_share / alen))
    sat = consensus[-1]
    for _ in range(n_arr):
        if L <= 2 * alen:
            break
        p = int(rng.integers(0, L - alen))
        arr = sat.repeat(alen // mono + 1)[:alen].clone()
        mut = torch.rand(alen, generator=g, device=device) < sdiv
        arr[mut] = torch.randint(0, 4, (int(mut.sum()),), generator=g, device=device, dtype=torch.uint8)
        ids[p:p + alen] = arr


def make_genome(params, device, scale=1.0, seed=1, n_contigs=24, repeat_copies=40000, repeat_len=300, repeat_div=0.12, realistic=False):
    """realistic=True: instead of the single planted family, the repeat landscape of REALISTIC_FAMILIES (about 45 % of every contig)."""
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    rng = np.random.default_rng(seed)
    sizes = [max(20000, int(s * scale)) for s in HG38_SIZES[:n_contigs]]
    al, rv = _codes(params)
    al_t = torch.tensor(al, dtype=torch.uint8, device=device)
    rv_t = torch.tensor(rv, dtype=torch.uint8, device=device)
    family = torch.randint(0, 4, (repeat_len,), generator=g, device=device, dtype=torch.uint8)
    consensus = None
    if realistic:
        consensus = [torch.randint(0, 4, (f[1],), generator=g, device=device, dtype=torch.uint8) for f in REALISTIC_FAMILIES]
        consensus.append(torch.randint(0, 4, (REALISTIC_SATELLITE[0],), generator=g, device=device, dtype=torch.uint8))
    G = Genome()
    G.names, G.sizes, G.ids, G.nmask_runs = NAMES[:n_contigs], sizes, [], []
    fw_words, rc_words, blocks = [], [], []
    anchors = [MARGIN * 32]
    s_words = 0
    per_contig_rep = max(1, int(repeat_copies * scale) // n_contigs)
    for ci, L in enumerate(sizes):
        ids = torch.randint(0, 4, (L,), generator=g, device=device, dtype=torch.uint8)
        if realistic:
            _plant_realistic(ids, L, 1.0, g, rng, device, consensus)
        # planted repeat family, diverged copies
        elif per_contig_rep and L > 4 * repeat_len:
            starts = torch.from_numpy(rng.integers(0, L - repeat_len, size=per_contig_rep)).to(device)
            idx = (starts[:, None] + torch.arange(repeat_len, device=device)[None, :]).reshape(-1)
            unit = family.repeat(per_contig_rep)
            mut = torch.rand(unit.shape, generator=g, device=device) < repeat_div
            unit = torch.where(mut, torch.randint(0, 4, unit.shape, generator=g, device=device, dtype=torch.uint8), unit)
            ids[idx] = unit
        # N runs: one large block in the middle, a few short gaps
        runs = []
        if L > 2_000_000:
            c0 = L // 2 - L // 100
            runs.append((c0, c0 + L // 50))
            for k in range(4):
                p = int(rng.integers(L // 10, L - L // 10))
                runs.append((p, p + 50_000))
        elif L > 100_000:
            runs.append((L // 2, L // 2 + 1000))
        runs = sorted(runs)
        merged = []
        for a, b in runs:
            if merged and a <= merged[-1][1]:
                merged[-1] = (merged[-1][0], max(merged[-1][1], b))
            else:
                merged.append((a, b))
        isn = torch.zeros(L, dtype=torch.bool, device=device)
        for a, b in merged:
            isn[a:b] = True
        n = (L + 31) // 32 + 2
        tot = n * 32
        code_f = torch.zeros(tot, dtype=torch.uint8, device=device)
        code_f[:L] = torch.where(isn, torch.zeros_like(ids), al_t[ids.long()])
        code_r = torch.zeros(tot, dtype=torch.uint8, device=device)
        code_r[tot - L:] = torch.where(isn, torch.zeros_like(ids), rv_t[ids.long()]).flip(0)
        fw_words.append(_pack(code_f))
        rc_words.append(_pack(code_r))
        # unmasked blocks (refbase.cpp:103-128) and their reverse-strand twins
        prev = 0
        for a, b in merged + [(L, L)]:
            if a - prev >= 16:
                blocks.append((2 * ci, prev, a))
                blocks.append((2 * ci + 1, tot - a, tot - prev))
            prev = b
        G.ids.append(ids)
        G.nmask_runs.append(merged)
        s_words += n
        anchors.append((s_words + MARGIN) * 32)
        del code_f, code_r, isn
    pad = torch.zeros(MARGIN, dtype=torch.int64, device=device)
    G.words = [torch.cat([pad] + fw_words + [pad]), torch.cat([pad] + rc_words + [pad])]
    G.anchors = np.array(anchors, dtype=np.uint32)
    G.rc_offsets = np.array([((L + 31) // 32 + 2) * 32 for L in sizes], dtype=np.uint32)
    blocks.sort(key=lambda t: (t[0], t[1]))
    G.blocks = np.array(blocks, dtype=np.uint32).reshape(-1, 3)
    return G


def genome_from_reference(params, ref, device):
    """A Genome (what make_reads / bench.py need) from a reference the PRODUCT's loader read (basal_amd.Reference on a FASTA, e.g. hg38):
    packed words of both strands as loaded, per-contig base ids decoded from the forward strand for the read sampler, N runs from the gaps
    between the loader's blocks (refbase.cpp:103-128: maximal runs without N of >= 16 bases)."""
    G = Genome()
    G.names = ref.names()
    G.sizes = [int(x) for x in ref.sizes()]
    G.anchors = np.array(ref.anchors(), dtype=np.uint32, copy=True)  # (copies: the arrays the loader hands out live as long as `ref` does)
    G.rc_offsets = np.array(ref.rc_offsets(), dtype=np.uint32, copy=True)
    G.blocks = np.array(ref.blocks(), dtype=np.uint32, copy=True)
    G.words = [torch.from_numpy(np.ascontiguousarray(ref.words(s)).view(np.int64).copy()).to(device) for s in (0, 1)]
    al, _ = _codes(params)
    inv = torch.zeros(4, dtype=torch.uint8)
    for base, code in enumerate(al):
        inv[code] = base
    inv = inv.to(device)
    sh = torch.arange(31, -1, -1, device=device, dtype=torch.int64) * 2
    G.ids, G.nmask_runs = [], []
    blocks = G.blocks.reshape(-1, 3)
    for ci, size in enumerate(G.sizes):
        w0 = int(G.anchors[ci]) // 32
        nw = (size + 31) // 32
        codes = ((G.words[0][w0:w0 + nw, None] >> sh[None, :]) & 3).reshape(-1)[:size]
        G.ids.append(inv[codes])
        runs, prev = [], 0
        for _, b, e in sorted((int(x[0]), int(x[1]), int(x[2])) for x in blocks[blocks[:, 0] == 2 * ci]):
            if b > prev:
                runs.append((prev, b))
            prev = e
        if prev < size:
            runs.append((prev, size))
        G.nmask_runs.append(runs)
    return G


def make_transcriptome(params, device, n_contigs=100_000, seed=1, median=1200, sigma=0.8, lo=300, hi=20_000):
    """A transcriptome-shaped reference for BASELINE.json config 3 ("transcriptome stand-in with <= 131 071
    contigs"): n_contigs sequences of log-normal length (median 1.2 kb, clipped to lo..hi), uniform ACGT, no N's -- about 0.15 Gbp at
    100 000 contigs.  Same HBM layout as make_genome (every contig starts on a word boundary and ends with two pad words, the reverse
    strand in the same slot), built without a Python loop over contigs; G.ids is ONE flat tensor with G.base_off, for make_pairs."""
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    rng = np.random.default_rng(seed)
    sizes = np.clip(rng.lognormal(np.log(median), sigma, n_contigs), lo, hi).astype(np.int64)
    nwords = (sizes + 31) // 32 + 2
    word_base = np.concatenate([[0], np.cumsum(nwords)])
    base_off = np.concatenate([[0], np.cumsum(sizes)])
    total = int(base_off[-1])
    al, rv = _codes(params)
    al_t = torch.tensor(al, dtype=torch.uint8, device=device)
    rv_t = torch.tensor(rv, dtype=torch.uint8, device=device)
    ids = torch.randint(0, 4, (total,), generator=g, device=device, dtype=torch.uint8)
    sizes_t = torch.from_numpy(sizes).to(device)
    cidx = torch.repeat_interleave(torch.arange(n_contigs, device=device), sizes_t)
    local = torch.arange(total, device=device) - torch.from_numpy(base_off[:-1]).to(device)[cidx]
    slot0 = torch.from_numpy(word_base[:-1] * 32).to(device)[cidx]
    slot_len = torch.from_numpy(nwords * 32).to(device)[cidx]
    tot_bits = int(word_base[-1]) * 32
    code_f = torch.zeros(tot_bits, dtype=torch.uint8, device=device)
    code_f[slot0 + local] = al_t[ids.long()]
    code_r = torch.zeros(tot_bits, dtype=torch.uint8, device=device)
    code_r[slot0 + slot_len - 1 - local] = rv_t[ids.long()]  # the reverse complement ends where the slot ends
    del cidx, local, slot0, slot_len
    pad = torch.zeros(MARGIN, dtype=torch.int64, device=device)
    G = Genome()
    G.flat = True
    G.names = ["t%06d" % i for i in range(n_contigs)]
    G.sizes = [int(x) for x in sizes]
    G.ids = ids
    G.base_off = torch.from_numpy(base_off[:-1]).to(device)
    G.nmask_runs = None
    G.words = [torch.cat([pad, _pack(code_f), pad]), torch.cat([pad, _pack(code_r), pad])]
    del code_f, code_r
    G.anchors = ((word_base + MARGIN) * 32).astype(np.uint32)
    G.rc_offsets = (nwords * 32).astype(np.uint32)
    blk = np.empty((2 * n_contigs, 3), np.uint32)  # one unmasked block per strand (refbase.cpp:103-128), in (id, begin) order
    blk[0::2, 0] = 2 * np.arange(n_contigs); blk[0::2, 1] = 0; blk[0::2, 2] = sizes
    blk[1::2, 0] = 2 * np.arange(n_contigs) + 1; blk[1::2, 1] = nwords * 32 - sizes; blk[1::2, 2] = nwords * 32
    G.blocks = blk
    return G


def make_reads(G, n, device, read_len=100, seed=2, p_conv=0.95, sub_rate=0.01, rev_frac=0.5, conv_from=1, conv_to=3, indel_frac=0.0, indel_max=2,
               del_base=None, del_frac=0.0, del_lo=20, del_hi=80):
    """n reads as ASCII bytes (n*read_len uint8 tensor) plus the truth (contig, 0-based start, strand).
    conv_to: one base id, or a list of ids (each converted base becomes one of them, uniformly: the multi-way rules).
    indel_frac: share of reads with one insertion or deletion of 1..indel_max bases.
    del_base / del_frac: share of reads with ONE base of that kind deleted between read positions del_lo and del_hi (config 5, BID-seq)."""
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    out_len = read_len
    pad = indel_max + 1 if (indel_frac > 0 or del_frac > 0) else 0
    read_len = out_len + pad  # sample a longer window, cut the read out of it below
    # sample (contig, start) uniformly over non-N stretches long enough for a read
    segs = []
    for ci, runs in enumerate(G.nmask_runs):
        prev = 0
        for a, b in runs + [(G.sizes[ci], G.sizes[ci])]:
            if a - prev > read_len + 1:
                segs.append((ci, prev, a - read_len))
            prev = b
    seg_len = torch.tensor([e - s for _, s, e in segs], dtype=torch.float64)
    pick = torch.multinomial(seg_len / seg_len.sum(), n, replacement=True, generator=torch.Generator().manual_seed(seed)).to(device)
    seg_ci = torch.tensor([c for c, _, _ in segs], device=device)[pick]
    seg_s = torch.tensor([s for _, s, _ in segs], device=device, dtype=torch.int64)[pick]
    seg_e = torch.tensor([e for _, _, e in segs], device=device, dtype=torch.int64)[pick]
    start = seg_s + (torch.rand(n, generator=g, device=device, dtype=torch.float64) * (seg_e - seg_s).double()).long()
    out = torch.empty((n, read_len), dtype=torch.uint8, device=device)
    ar = torch.arange(read_len, device=device)
    for ci in range(len(G.sizes)):
        m = (seg_ci == ci).nonzero(as_tuple=True)[0]
        if m.numel() == 0:
            continue
        out[m] = G.ids[ci][(start[m][:, None] + ar[None, :])]
    rev = torch.rand(n, generator=g, device=device) < rev_frac
    out = torch.where(rev[:, None], (3 - out).flip(1), out)
    if pad:  # one gap per chosen read, in read orientation: deletion = skip k window bases, insertion = k random bases
        L = out_len
        col = torch.arange(L, device=device)[None, :]
        k = torch.zeros(n, dtype=torch.int64, device=device)
        pos = torch.zeros(n, dtype=torch.int64, device=device)
        ins = torch.zeros(n, dtype=torch.bool, device=device)
        if indel_frac > 0:
            has = torch.rand(n, generator=g, device=device) < indel_frac
            kk = 1 + (torch.rand(n, generator=g, device=device) * indel_max).long().clamp(max=indel_max - 1)
            k = torch.where(has, kk, k)
            pos = torch.where(has, 10 + (torch.rand(n, generator=g, device=device) * (L - 20)).long(), pos)
            ins = has & (torch.rand(n, generator=g, device=device) < 0.5)
        if del_frac > 0:
            want = torch.rand(n, generator=g, device=device) < del_frac
            p0 = del_lo + (torch.rand(n, generator=g, device=device) * (del_hi - del_lo)).long()
            cand = (out[:, :L] == del_base) & (col >= p0[:, None]) & (col < del_hi)  # the first such base at or behind p0
            first = torch.where(cand.any(dim=1), cand.float().argmax(dim=1), torch.full_like(p0, -1))
            ok = want & (first >= 0)
            k = torch.where(ok, torch.ones_like(k), k)
            pos = torch.where(ok, first, pos)
            ins = ins & ~ok
        dele = (k > 0) & ~ins
        idx = col + torch.where(dele[:, None] & (col >= pos[:, None]), k[:, None], torch.zeros_like(col))
        idx = idx - torch.where(ins[:, None] & (col >= (pos + k)[:, None]), k[:, None], torch.zeros_like(col))
        cut = torch.gather(out, 1, idx.clamp(0, read_len - 1))
        in_ins = ins[:, None] & (col >= pos[:, None]) & (col < (pos + k)[:, None])
        cut = torch.where(in_ins, torch.randint(0, 4, cut.shape, generator=g, device=device, dtype=torch.uint8), cut)
        out = cut
        read_len = out_len
    tos = list(conv_to) if isinstance(conv_to, (list, tuple)) else [conv_to]
    conv = (out == conv_from) & (torch.rand(out.shape, generator=g, device=device) < p_conv)
    if len(tos) == 1:
        out = torch.where(conv, torch.full_like(out, tos[0]), out)
    else:
        pick_to = torch.tensor(tos, dtype=torch.uint8, device=device)[torch.randint(0, len(tos), out.shape, generator=g, device=device)]
        out = torch.where(conv, pick_to, out)
    sub = torch.rand(out.shape, generator=g, device=device) < sub_rate
    out = torch.where(sub, (out + torch.randint(1, 4, out.shape, generator=g, device=device, dtype=torch.uint8)) % 4, out)
    ascii_t = torch.tensor([65, 67, 71, 84], dtype=torch.uint8, device=device)
    bases = ascii_t[out.long()].reshape(-1).contiguous()
    return bases, seg_ci, start, rev


def make_pairs(G, n, device, read_len=150, frag_min=200, frag_max=600, seed=3, p_conv=0.9, sub_rate=0.01, rev_frac=0.5, conv_from=0, conv_to=2):
    """n read pairs (two n*read_len uint8 ASCII tensors): fragments of frag_min..frag_max bases, mate 1 = the fragment's first read_len
    bases, mate 2 = the reverse complement of its last read_len bases; the conversion acts on the fragment's mate-1 strand, so mate 2
    shows its complement."""
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    if getattr(G, "flat", False):  # make_transcriptome: one flat base array, fragments from the contigs long enough to hold one
        sizes_t = torch.tensor(G.sizes, dtype=torch.int64)
        room = (sizes_t - frag_max).clamp(min=0).double()
        pick = torch.multinomial(room / room.sum(), n, replacement=True, generator=torch.Generator().manual_seed(seed)).to(device)
        room_p = room.to(device)[pick]
        start = G.base_off[pick] + (torch.rand(n, generator=g, device=device, dtype=torch.float64) * room_p).long()
        flen = frag_min + (torch.rand(n, generator=g, device=device) * (frag_max - frag_min + 1)).long().clamp(max=frag_max - frag_min)
        flen = torch.maximum(flen, torch.full_like(flen, read_len))
        ar = torch.arange(read_len, device=device)
        a_ = G.ids[start[:, None] + ar[None, :]]
        b_ = 3 - G.ids[(start + flen - 1)[:, None] - ar[None, :]]
        return _finish_pairs(a_, b_, n, g, device, p_conv, sub_rate, rev_frac, conv_from, conv_to)
    segs = []
    for ci, runs in enumerate(G.nmask_runs):
        prev = 0
        for a, b in runs + [(G.sizes[ci], G.sizes[ci])]:
            if a - prev > frag_max + 1:
                segs.append((ci, prev, a - frag_max))
            prev = b
    seg_len = torch.tensor([e - s for _, s, e in segs], dtype=torch.float64)
    pick = torch.multinomial(seg_len / seg_len.sum(), n, replacement=True, generator=torch.Generator().manual_seed(seed)).to(device)
    seg_ci = torch.tensor([c for c, _, _ in segs], device=device)[pick]
    seg_s = torch.tensor([s for _, s, _ in segs], device=device, dtype=torch.int64)[pick]
    seg_e = torch.tensor([e for _, _, e in segs], device=device, dtype=torch.int64)[pick]
    start = seg_s + (torch.rand(n, generator=g, device=device, dtype=torch.float64) * (seg_e - seg_s).double()).long()
    flen = frag_min + (torch.rand(n, generator=g, device=device) * (frag_max - frag_min + 1)).long().clamp(max=frag_max - frag_min)
    flen = torch.maximum(flen, torch.full_like(flen, read_len))
    a_ = torch.empty((n, read_len), dtype=torch.uint8, device=device)
    b_ = torch.empty((n, read_len), dtype=torch.uint8, device=device)
    ar = torch.arange(read_len, device=device)
    for ci in range(len(G.sizes)):
        m = (seg_ci == ci).nonzero(as_tuple=True)[0]
        if m.numel() == 0:
            continue
        a_[m] = G.ids[ci][start[m][:, None] + ar[None, :]]
        b_[m] = 3 - G.ids[ci][(start[m] + flen[m] - 1)[:, None] - ar[None, :]]
    return _finish_pairs(a_, b_, n, g, device, p_conv, sub_rate, rev_frac, conv_from, conv_to)


def _finish_pairs(a_, b_, n, g, device, p_conv, sub_rate, rev_frac, conv_from, conv_to):
    """a_ = the fragment's first bases, b_ = the reverse complement of its last ones: orientation, conversion, substitutions, ASCII."""
    rev = torch.rand(n, generator=g, device=device) < rev_frac
    r1 = torch.where(rev[:, None], b_, a_)
    r2 = torch.where(rev[:, None], a_, b_)
    c1 = (r1 == conv_from) & (torch.rand(r1.shape, generator=g, device=device) < p_conv)
    r1 = torch.where(c1, torch.full_like(r1, conv_to), r1)
    c2 = (r2 == 3 - conv_from) & (torch.rand(r2.shape, generator=g, device=device) < p_conv)
    r2 = torch.where(c2, torch.full_like(r2, 3 - conv_to), r2)
    for r in (r1, r2):
        sub = torch.rand(r.shape, generator=g, device=device) < sub_rate
        r[sub] = ((r + torch.randint(1, 4, r.shape, generator=g, device=device, dtype=torch.uint8)) % 4)[sub]
    ascii_t = torch.tensor([65, 67, 71, 84], dtype=torch.uint8, device=device)
    return ascii_t[r1.long()].reshape(-1).contiguous(), ascii_t[r2.long()].reshape(-1).contiguous()
