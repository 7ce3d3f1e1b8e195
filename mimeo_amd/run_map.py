"""`mimeo map` — all high-identity segments shared between genomes (reference:
src/mimeo/run_map.py:190-328)."""
import argparse
import logging
import os
import sys

from . import _cli, engine, formats, workflow


def mainArgs(argv=None):
    parser = argparse.ArgumentParser(
        description='Find all high-identity segments shared between genomes.', prog='mimeo-map')
    _cli.add_common(parser, 'mimeo-map', None, 'BHit', 'BHit', with_b=True)
    parser.add_argument('--TRFpath', type=str, default='trf', help='Accepted for compatibility; TRF is not used.')
    for name, default, what in (('tmatch', 2, 'matching weight'), ('tmismatch', 7, 'mismatching penalty'), ('tdelta', 7, 'indel penalty'),
                                ('tPM', 80, 'match probability: accepted for compatibility, the on-GPU scorer tries every period (no detection statistics)'),
                                ('tPI', 10, 'indel probability: accepted for compatibility (as --tPM)'),
                                ('tminscore', 50, 'minimum alignment score of a tandem repeat'), ('tmaxperiod', 50, 'maximum period size (<= 64)')):
        parser.add_argument('--' + name, type=int, default=default, help='Tandem scorer (TRF parameter of the same name): %s.' % what)
    parser.add_argument('--maxtandem', type=float, default=None,
                        help='Max percentage of an A-genome alignment which may be masked by TRF.')
    parser.add_argument('--writeTRF', action='store_true', default=False, help='Write TRF-filtered alignment file.')
    return parser.parse_args(argv)


def main(argv=None):
    args = mainArgs(argv)
    dist, outdir = _cli.start(args)
    logging.info('Starting genome mapping workflow.')
    A = _cli.load_genome(args.afasta, args.adir, 'A')
    B = _cli.load_genome(args.bfasta, args.bdir, 'B')
    outtab = os.path.join(outdir, args.outfile)
    chrLens = formats.chromlens(A.names, A.lengths)  # run_map.py:255 (no file)
    pairs = workflow.all_pairs(len(A.names), len(B.names))
    if not pairs:
        logging.error('No files to align. Check --adir and --bdir contain at least one fasta each.')
        sys.exit(1)
    workflow.map_hits(A, B, pairs, outtab, minIdt=args.minIdt, minLen=args.minLen, hspthresh=args.hspthresh,
                      reuseTab=args.recycle, dist=dist)
    if dist.rank == 0:
        logging.info('Importing alignments from %s' % outtab)
        rows = formats.parse_tab(outtab)
        if not rows:
            logging.warning('No alignments found in %s' % outtab)
            sys.exit(1)
        rows = formats.import_align(rows, prefix=args.prefix, min_len=args.minLen, min_idt=args.minIdt)
        if args.maxtandem:  # run_map.py:293-314: tandem filter, optional .trf table
            logging.info('Filtering alignments by tandem repeat content...')
            rows = workflow.trf_filter(rows, A, prefix=args.prefix, tmatch=args.tmatch, tmismatch=args.tmismatch,
                                       tminscore=args.tminscore, tmaxperiod=args.tmaxperiod, maxtandem=args.maxtandem, tdelta=args.tdelta)
            if args.writeTRF:
                logging.info('Writing TRF-filtered alignments to file: %s' % (outtab + '.trf'))
                formats.write_trf_tab(rows, outtab)
        if args.gffout:
            gffout = os.path.join(outdir, args.gffout)
            logging.info('Writing GFF3 output to %s' % gffout)
            with open(gffout, 'w') as f:
                for x in formats.gff_map_lines(rows, chrlens=chrLens, ftype=args.label):
                    f.write(x)
    if args.verbose:
        logging.info('engine stats: %s', engine.stats())
    A.close()
    B.close()
    logging.info('Finished!')


if __name__ == '__main__':
    main()
