"""`mimeo self` — internal repeat finder (reference: src/mimeo/run_self.py:169-255)."""
import argparse
import logging
import os

from . import _cli, engine, formats, workflow


def mainArgs(argv=None):
    parser = argparse.ArgumentParser(
        description='Internal repeat finder. Mimeo-self aligns a genome to itself and extracts high-identity '
                    'segments above an coverage threshold.', prog='mimeo-self')
    _cli.add_common(parser, 'mimeo-self', 'mimeo-self_repeats.gff3', 'Self_Repeat', 'Self_Repeat', with_b=False)
    parser.add_argument('--bedtools', type=str, default='bedtools', help='Accepted for compatibility; bedtools is not used.')
    parser.add_argument('--minCov', type=int, default=3, help='Minimum depth of aligned segments to report repeat feature.')
    parser.add_argument('--intraCov', type=int, default=5,
                        help='Minimum depth of aligned segments from same scaffold to report feature. Used if "--strictSelf" mode is selected.')
    parser.add_argument('--strictSelf', action='store_true',
                        help='If set process same-scaffold alignments separately with option to use higher "--intraCov" threshold.')
    return parser.parse_args(argv)


def main(argv=None):
    args = mainArgs(argv)
    dist, outdir = _cli.start(args)
    logging.info('Starting self-alignment workflow.')
    A = _cli.load_genome(args.afasta, args.adir, 'A')
    outtab = os.path.join(outdir, args.outfile)
    gffout = os.path.join(outdir, args.gffout)
    if dist.rank == 0:
        formats.chromlens(A.names, A.lengths, os.path.join(outdir, 'A_gen_lens.txt'))  # run_self.py:223-224
    pairs = workflow.all_pairs(len(A.names))
    logging.info('Running alignments...')
    workflow.self_repeats(A, pairs, outtab, gffout, minIdt=args.minIdt, minLen=args.minLen, hspthresh=args.hspthresh,
                          minCov=args.minCov, intraCov=args.intraCov, splitSelf=args.strictSelf, reuseTab=args.recycle,
                          label=args.label, prefix=args.prefix, dist=dist)
    if args.verbose:
        logging.info('engine stats: %s', engine.stats())
    A.close()
    logging.info('Finished!')


if __name__ == '__main__':
    main()
