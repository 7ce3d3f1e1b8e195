#!/usr/bin/env python3
"""Opcode mix of the seed-scan kernel K34 from its ISA, priced with the measured per-opcode issue rates: what the
`roofline.valu` block of the bench line is computed from.

    python scripts/k34_isa_mix.py [tag]      -> profiles/<tag>_k34_isa_mix.json   (needs hipcc; run in the build container)

tag r03b (default): the level form of the first pass, compiled alone (-DK34_ONLY_FORM=0: the shipped kernel also holds the
lane-major form behind a switch, whose blocks never run), counters from profiles/r03b_pmc_k34_sq.json; tag r03: the lane-major
form (-DK34_ONLY_FORM=96) with profiles/r03_pmc_k34_sq.json — the mix behind profiles/r03_bench_c4_rows_line.json.

Inputs
  * the device assembly of mimeo_amd/csrc/k34_fused.hip (hipcc -S --cuda-device-only, the Makefile's flags), first-pass
    kernel k34_scan_extend<512, 1280, false>;
  * profiles/r03_valu_rate.txt (scripts/ubench/valu_rate.hip on MI355X): nanoseconds per wavefront-instruction and SIMD
    at 8 waves per SIMD, quoted as cycles at the nominal 2.4 GHz.  Two classes fall out of it: simple VOP2 integer ops and
    v_add_f32 (add / sub / and / or / xor / shifts / mov) at ~2.55 cycles, everything else (VOP3, compares, selects, max /
    min, 24-bit multiplies, popcounts, funnel shifts, DPP, SDWA, packed ops) at ~4.35;
  * profiles/r03_pmc_k34_sq.json: SQ_INSTS_VALU per first-pass launch on a C4 unit (10 Mbp x 10 Mbp).
The kernel's work is two nested bodies: the PAIR ROUND (64 seed hits: descriptor read, 13 ds_bpermute + 3 ds_read_b128,
the pre-filter, compaction) and the CHUNK VISIT (64 target entries against one query segment: 13 probes in LDS, prefix
sum, descriptor emission).  Static instruction counts per body come from the basic blocks between the loop headers the
compiler names; the dynamic weights are the launch's rounds (seed hits / 64 / lane utilisation of the rounds, measured
by SQ_THREAD_CYCLES_VALU / SQ_INSTS_VALU) and visits (tiles x segments x chunks), and the model's total is checked
against the measured SQ_INSTS_VALU.
"""
import collections
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KERNEL = '_ZN5mimeo15k34_scan_extendILi512ELj1280ELb0EEEvNS_9FusedArgsE'
FULL_RATE = {'v_add_u32', 'v_sub_u32', 'v_subrev_u32', 'v_xor_b32', 'v_or_b32', 'v_and_b32', 'v_lshrrev_b32', 'v_lshlrev_b32',
             'v_ashrrev_i32', 'v_mov_b32', 'v_not_b32', 'v_add_f32', 'v_sub_f32', 'v_add_co_u32', 'v_sub_co_u32', 'v_addc_co_u32',
             'v_subb_co_u32', 'v_xnor_b32'}


def rates():
    """class -> cycles at 2.4 GHz per wavefront-instruction and SIMD, averaged over the ubench rows of the class"""
    full, half = [], []
    with open(os.path.join(ROOT, 'profiles', 'r03_valu_rate.txt')) as f:
        for line in f:
            m = re.match(r'^(\S.*?)\s+waves/SIMD 8\s.*= ([0-9.]+) cycles', line)
            if not m:
                continue
            name, cyc = m.group(1), float(m.group(2))
            if cyc > 8:          # ds_bpermute (LDS crossbar) and the vcc-chained select: not VALU issue rates
                continue
            op = name.split()[0]
            if (op.split('/')[0] in FULL_RATE or name.startswith('v_xor/or/sub/ashr') or name.startswith('v_lshrrev / v_and')) and 'dpp' not in name and 'sdwa' not in name:
                full.append(cyc)
            elif not name.startswith('v_pk_fma') and not name.startswith('v_fma') and not name.startswith('v_fmac'):
                half.append(cyc)
    return sum(full) / len(full), sum(half) / len(half), len(full), len(half)


def classify(op):
    base = re.sub(r'_(e32|e64|dpp|sdwa)$', '', op)
    if op.startswith('v_'):
        if base in FULL_RATE and not op.endswith('_dpp') and not op.endswith('_sdwa'):
            return 'valu_full'
        return 'valu_half'
    if op.startswith('s_'):
        return 'salu'
    if op.startswith('ds_'):
        return 'lds'
    if op.startswith(('global_', 'buffer_', 'flat_')):
        return 'vmem'
    return 'other'


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else 'r03b'
    form = '96' if tag == 'r03' else '0'
    with tempfile.TemporaryDirectory() as td:
        asm = os.path.join(td, 'k34.s')
        subprocess.check_call(['/opt/rocm/bin/hipcc', '-O3', '-std=c++17', '--offload-arch=gfx950', '-ffp-contract=off', '-w', '-S', '-DK34_ONLY_FORM=' + form,
                               '--cuda-device-only', '-o', asm, os.path.join(ROOT, 'mimeo_amd', 'csrc', 'k34_fused.hip')],
                              stderr=subprocess.DEVNULL)
        src = open(asm).read().split('\n')
    start = next(i for i, l in enumerate(src) if l.startswith(KERNEL + ':'))
    end = next(i for i in range(start, len(src)) if src[i].startswith('.Lfunc_end'))
    blocks, cur = [], None
    depth_of = {}
    for l in src[start:end]:
        s = l.strip()
        m = re.match(r'^(\.LBB\S+):\s*(;.*)?$', s)
        if m:
            cur = [m.group(1), []]
            blocks.append(cur)
            d = re.findall(r'Depth=(\d+)', s)
            depth_of[m.group(1)] = max(map(int, d)) if d else 0
            continue
        if s.startswith(';') and cur is not None and not cur[1]:   # a loop header's comment runs over several lines: the deepest counts
            d = re.findall(r'Depth=(\d+)', s)
            if d:
                depth_of[cur[0]] = max(depth_of[cur[0]], max(map(int, d)))
            continue
        if not s or s.startswith(';') or s.startswith('.'):
            continue
        if cur is None:
            cur = ['entry', []]
            blocks.append(cur)
            depth_of['entry'] = 0
        cur[1].append(s.split()[0])
    per_block = []
    for name, ops in blocks:
        c = collections.Counter(classify(o) for o in ops)
        per_block.append({'block': name, 'depth': depth_of.get(name, 0), 'n': len(ops), **{k: c.get(k, 0) for k in ('valu_full', 'valu_half', 'salu', 'lds', 'vmem')},
                          'bpermute': sum(1 for o in ops if o == 'ds_bpermute_b32'), 'bcnt': sum(1 for o in ops if o.startswith('v_bcnt'))})
    # the pair round: from the block that issues the 13 ds_bpermute of a frame fetch to the end of the depth-4 loop that holds it;
    # the pre-filter is the block with the popcounts
    fetch = next(i for i, b in enumerate(per_block) if b['bpermute'] >= 12 and b['depth'] >= 3)
    d4 = per_block[fetch]['depth']
    lo = fetch
    while lo > 0 and per_block[lo - 1]['depth'] >= d4:
        lo -= 1
    hi = fetch
    while hi + 1 < len(per_block) and per_block[hi + 1]['depth'] >= d4:
        hi += 1
    rnd = per_block[lo:hi + 1]
    filt = max(rnd, key=lambda b: b['bcnt'])
    # the chunk visit: the depth-2/3 blocks around it (everything inside the segment loop that is not the round loop)
    seg_depth = d4 - 2
    vlo = lo
    while vlo > 0 and per_block[vlo - 1]['depth'] >= seg_depth:
        vlo -= 1
    vhi = hi
    while vhi + 1 < len(per_block) and per_block[vhi + 1]['depth'] >= seg_depth:
        vhi += 1
    visit = [b for i, b in enumerate(per_block[vlo:vhi + 1], vlo) if not (lo <= i <= hi)]

    def tot(bs, k):
        return sum(b[k] for b in bs)

    c_full, c_half, n_full, n_half = rates()
    pmc = json.load(open(os.path.join(ROOT, 'profiles', tag + '_pmc_k34_sq.json')))['kernels']['k34_scan_extend (first pass)']
    insts = pmc['SQ_INSTS_VALU']
    lane_util = pmc['SQ_THREAD_CYCLES_VALU'] / insts / 64.0 if pmc.get('SQ_THREAD_CYCLES_VALU') else None
    # dynamic weights on a C4 unit: 7.78e7 seed hits; a round holds 64 descriptors but the last of a (wavefront, chunk visit) is
    # partly empty; visits = 4096 tiles x 2 segments x ceil(2441 / 64) chunks
    hits = 7.78e7
    visits = 4096 * 2 * 39
    rounds_full = hits / 64
    # straight-line share of a round actually executed: the filter block and the fetch always, the flush / emission side blocks rarely
    round_valu = {'valu_full': filt['valu_full'] + per_block[fetch]['valu_full'], 'valu_half': filt['valu_half'] + per_block[fetch]['valu_half']}
    round_all = {'valu_full': tot(rnd, 'valu_full'), 'valu_half': tot(rnd, 'valu_half')}
    visit_all = {'valu_full': tot(visit, 'valu_full'), 'valu_half': tot(visit, 'valu_half')}
    if form == '96':
        # lane-major form.  rounds: partial last rounds — about half a round per visit on top of the full ones; a visit executes its
        # probe block once and the emission loop body ~3.4 times (non-empty probes per lane, worst lane of 64: ~9)
        rounds = rounds_full + 0.5 * visits
        model = rounds * (round_valu['valu_full'] + round_valu['valu_half']) + visits * (visit_all['valu_full'] + visit_all['valu_half']) * 1.6
    else:
        # level form.  The visit's last, partial round is a copy of the round among the visit's own blocks; the level loop body
        # (~9 VALU) runs ~36 times in a visit of the segment's own half of the key space (12 probes x ~3 levels) and ~3 times in
        # a visit of the other half: ~19.5 x 9 / 370 = +0.47 of the visit's static count
        rounds = rounds_full
        model = rounds * (round_valu['valu_full'] + round_valu['valu_half']) + visits * (visit_all['valu_full'] + visit_all['valu_half']) * 1.47
    share_round = rounds * (round_valu['valu_full'] + round_valu['valu_half']) / insts
    full_share = (rounds * round_valu['valu_full'] + (insts - rounds * (round_valu['valu_full'] + round_valu['valu_half'])) *
                  visit_all['valu_full'] / max(1, visit_all['valu_full'] + visit_all['valu_half'])) / insts
    c_mix = full_share * c_full + (1 - full_share) * c_half
    out = {
        'what': 'opcode mix of k34_scan_extend<512,1280,false> (first pass, %s form compiled alone) from its gfx950 ISA, priced with profiles/r03_valu_rate.txt' % ('lane-major' if form == '96' else 'level'),
        'issue_cycles_at_2.4GHz': {'full_rate_class': round(c_full, 3), 'half_rate_class': round(c_half, 3), 'ubench_rows': [n_full, n_half],
                                   'full_rate_ops': sorted(FULL_RATE)},
        'static': {'pair_round_blocks': rnd, 'prefilter_block': filt, 'chunk_visit_blocks': visit,
                   'pair_round_straight_line': round_valu, 'pair_round_all_blocks': round_all, 'chunk_visit_all_blocks': visit_all},
        'dynamic_c4_unit': {'seed_hits': hits, 'rounds': rounds, 'visits': visits, 'valu_insts_measured': insts, 'valu_insts_model': round(model),
                            'share_of_valu_in_pair_rounds': round(share_round, 3), 'full_rate_share_of_valu': round(full_share, 3),
                            'lanes_active_per_valu_inst': round(lane_util, 3) if lane_util else None},
        'cycles_per_inst_mix': round(c_mix, 3),
        'note': 'frac_of_issue_peak = valu_insts x cycles_per_inst_mix / (1024 SIMDs x launch seconds x 2.4e9): the ubench cycles and the kernel '
                'time are both wall-clock based, so the nominal 2.4 GHz cancels.  SQ_ACTIVE_INST_VALU counts one quad-cycle per VALU '
                'instruction whatever its rate (profiles/r03_pmc_k34_sq.json: ratio 1.000), so it cannot tell the two classes apart.',
    }
    path = os.path.join(ROOT, 'profiles', tag + '_k34_isa_mix.json')
    json.dump(out, open(path, 'w'), indent=1)
    print(json.dumps({k: out[k] for k in ('issue_cycles_at_2.4GHz', 'dynamic_c4_unit', 'cycles_per_inst_mix')}, indent=1))
    print('pair round: %d blocks, straight line %s; pre-filter block %s; chunk visit %d blocks %s' % (
        len(rnd), round_valu, {k: filt[k] for k in ('block', 'valu_full', 'valu_half', 'lds')}, len(visit), visit_all))


if __name__ == '__main__':
    main()
