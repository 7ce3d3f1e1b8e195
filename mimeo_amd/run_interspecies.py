"""`mimeo x` — regions of genome A covered by many segments of genome B (reference:
src/mimeo/run_interspecies.py:173-258; hspthresh is parsed but the reference never forwards
it, so lastz always runs at 3000 there — reproduced)."""
import argparse
import logging
import os

from . import _cli, engine, formats, workflow


def mainArgs(argv=None):
    parser = argparse.ArgumentParser(
        description='Cross-species repeat finder. Mimeo-x searches for features which are abundant in an external genome.',
        prog='mimeo-x')
    _cli.add_common(parser, 'mimeo-x', 'mimeo_B_in_A.gff3', 'B_Repeat', 'B_Repeat', with_b=True)
    parser.add_argument('--bedtools', type=str, default='bedtools', help='Accepted for compatibility; bedtools is not used.')
    parser.add_argument('--minCov', type=int, default=5, help='Minimum depth of B-genome hits to report feature in A-genome.')
    return parser.parse_args(argv)


def main(argv=None):
    args = mainArgs(argv)
    dist, outdir = _cli.start(args)
    logging.info('Starting cross-species repeat identification...')
    A = _cli.load_genome(args.afasta, args.adir, 'A')
    B = _cli.load_genome(args.bfasta, args.bdir, 'B')
    outtab = os.path.join(outdir, args.outfile)
    gffout = os.path.join(outdir, args.gffout)
    if dist.rank == 0:
        formats.chromlens(A.names, A.lengths, os.path.join(outdir, 'A_gen_lens.txt'))
    pairs = workflow.all_pairs(len(A.names), len(B.names))
    logging.info('Running alignments...')
    workflow.self_repeats(A, pairs, outtab, gffout, minIdt=args.minIdt, minLen=args.minLen, hspthresh=3000,
                          minCov=args.minCov, reuseTab=args.recycle, label=args.label, prefix=args.prefix, dist=dist,
                          source='mimeo', B=B)
    if args.verbose:
        logging.info('engine stats: %s', engine.stats())
    A.close()
    B.close()
    logging.info('Finished!')


if __name__ == '__main__':
    main()
