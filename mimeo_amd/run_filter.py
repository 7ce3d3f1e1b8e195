"""`mimeo filter` — drop library sequences that are mostly tandem repeats (reference:
src/mimeo/run_filter.py:127-210 + wrappers.py:265-377 trfFasta).  The on-GPU tandem scorer (K8,
DESIGN.md "Tandem scorer v2") stands in for `trf ... -m -h -ngs`: a record stays when
(masked + N) / len * 100 < maxtandem, the test trfFasta applies to TRF's masked output
(wrappers.py:369-371).  PARITY UNPINNED against TRF itself (absent here)."""
import argparse
import logging
import os

import numpy as np

from . import _cli, engine, formats


def mainArgs(argv=None):
    parser = argparse.ArgumentParser(description='Filter SSR containing sequences from fasta library of repeats.',
                                     prog='mimeo-filter')
    parser.add_argument('--version', action='version', version='mimeo-filter %s' % _cli.__version__)
    parser.add_argument('--infile', type=str, required=True, help='Multifasta library of repeats to filter.')
    parser.add_argument('-d', '--outdir', type=str, default=None, help='Write output files to this directory. (Default: cwd)')
    parser.add_argument('--outfile', type=str, default=None, help='Name of the filtered FASTA file.')
    parser.add_argument('--keeptemp', action='store_true', default=False, help='Accepted for compatibility (no temp files are made).')
    parser.add_argument('--verbose', action='store_true', default=False, help='Report per-record masked fractions.')
    parser.add_argument('--TRFpath', type=str, default='trf', help='Accepted for compatibility; TRF is not used.')
    parser.add_argument('--tmatch', type=int, default=2, help='Tandem scorer matching weight')
    parser.add_argument('--tmismatch', type=int, default=7, help='Tandem scorer mismatching penalty')
    parser.add_argument('--tdelta', type=int, default=7, help='Tandem scorer indel penalty (0: gap-free comparison)')
    parser.add_argument('--tPM', type=int, default=80, help='Accepted for compatibility (no detection statistics: every period is tried).')
    parser.add_argument('--tPI', type=int, default=10, help='Accepted for compatibility.')
    parser.add_argument('--tminscore', type=int, default=50, help='Minimum tandem score to mask')
    parser.add_argument('--tmaxperiod', type=int, default=50, help='Maximum period size to score (<= 64).')
    parser.add_argument('--maxtandem', type=float, default=40,
                        help='Max percentage of a sequence which may be masked. If exceeded, element will be discarded.')
    parser.add_argument('--loglevel', type=str, default='INFO', choices=['DEBUG', 'INFO', 'WARNING', 'ERROR', 'CRITICAL'])
    parser.add_argument('--device', type=int, default=None, help='GPU index (default 0).')
    return parser.parse_args(argv)


def filter_fasta(infile, outfile, tmatch=2, tmismatch=7, tminscore=50, tmaxperiod=50, maxtandem=40, verbose=False, tdelta=7):
    """wrappers.py:265-377 trfFasta; returns the ids kept (in file order)."""
    G = engine.Genome.from_fasta(infile)
    headers = []
    names, seqs = formats.read_fasta(infile, headers)  # the text is needed again to write the survivors
    iv = np.array([(i, 0, ln) for i, ln in enumerate(G.lengths)], dtype=np.uint32).reshape(-1, 3)
    masked = engine.tandem_masked(G, iv, tmatch, tmismatch, tminscore, tmaxperiod, tdelta)
    G.close()
    keep = []
    for n, s, m in zip(names, seqs, masked.tolist()):
        if len(s) == 0:
            continue  # the reference divides by len(rec.seq); an empty record cannot pass
        # K8 counts masked ACGT-or-N positions of tandem segments; Ns outside them count as well, as
        # rec.seq.count('N') does on TRF's masked output (wrappers.py:369: upper-case N only, so a
        # soft-masked record with 'n' runs is not penalised for them)
        n_count = int((s == ord('N')).sum())
        pct = min(len(s), m + n_count) / len(s) * 100
        if verbose:
            logging.info('%s\tlen %d\tmasked %.1f%%', n, len(s), pct)
        if pct < float(maxtandem):
            keep.append(n)
    kept = set(keep)
    with open(outfile, 'wb') as f:
        for n, hd, s in zip(names, headers, seqs):
            if n in kept:
                f.write(b'>' + hd.encode() + b'\n')  # SeqIO.write emits '>id description' (wrappers.py:373-377)
                b = s.tobytes()
                for i in range(0, len(b), 60):
                    f.write(b[i:i + 60] + b'\n')
    return keep


def main(argv=None):
    args = mainArgs(argv)
    _cli.init_logging(args.loglevel)
    logging.info('Starting SSR filtering process.')
    engine.init(args.device if args.device is not None else int(os.environ.get('MIMEO_FORCE_DEVICE', 0)))
    outname = args.outfile or os.path.splitext(os.path.basename(args.infile))[0] + '_filtered.fa'  # run_filter.py:170-175
    outdir = os.path.abspath(args.outdir) if args.outdir else os.getcwd()
    os.makedirs(outdir, exist_ok=True)
    infile = os.path.abspath(args.infile)
    if not os.path.isfile(infile):
        logging.error('Input fasta not found at path: %s' % infile)
        raise SystemExit(1)
    keep = filter_fasta(infile, os.path.join(outdir, outname), args.tmatch, args.tmismatch, args.tminscore, args.tmaxperiod,
                        args.maxtandem, args.verbose, args.tdelta)
    logging.info('Kept %d sequences.' % len(keep))
    logging.info('Finished!')


if __name__ == '__main__':
    main()
