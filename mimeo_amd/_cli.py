"""Shared CLI plumbing for `mimeo self | x | map` (reference: src/mimeo/run_self.py:32-166,
run_interspecies.py:38-170, run_map.py:36-187 — same flag names, types and defaults)."""
import logging
import os
import sys

from . import engine, formats
from .dist import Dist

__version__ = '0.1.0+mi355x'


def add_common(parser, prog, gff_default, label, prefix, with_b):
    parser.add_argument('--version', action='version', version='%s %s' % (prog, __version__))
    parser.add_argument('--adir', type=str, default=None,
                        help='Directory containing sequences of genome A (one or more FASTA files).')
    if with_b:
        parser.add_argument('--bdir', type=str, default=None, help='Directory containing sequences of genome B.')
    parser.add_argument('--afasta', type=str, default=None, help='A genome as multifasta.')
    if with_b:
        parser.add_argument('--bfasta', type=str, default=None, help='B genome as multifasta.')
    parser.add_argument('-r', '--recycle', action='store_true', help='Use existing alignment "--outfile" if found.')
    parser.add_argument('-d', '--outdir', type=str, default=None, help='Write output files to this directory. (Default: cwd)')
    parser.add_argument('--gffout', type=str, default=gff_default, help='Name of GFF3 annotation file.')
    parser.add_argument('--outfile', type=str, default='mimeo_alignment.tab', help='Name of alignment result file.')
    parser.add_argument('--verbose', action='store_true', default=False, help='Report engine stage statistics.')
    parser.add_argument('--label', type=str, default=label, help='Set annotation TYPE field in gff.')
    parser.add_argument('--prefix', type=str, default=prefix, help='ID prefix for features.')
    parser.add_argument('--keeptemp', action='store_true', default=False, help='Accepted for compatibility (no temp files are made).')
    parser.add_argument('--lzpath', type=str, default='lastz', help='Accepted for compatibility; LASTZ is not used.')
    parser.add_argument('--minIdt', type=int, default=60, help='Minimum alignment identity to report.')
    parser.add_argument('--minLen', type=int, default=100, help='Minimum alignment length to report.')
    parser.add_argument('--hspthresh', type=int, default=3000, help='HSP min score threshold.')
    parser.add_argument('--loglevel', type=str, default='INFO', choices=['DEBUG', 'INFO', 'WARNING', 'ERROR', 'CRITICAL'],
                        help='Set the logging level.')
    parser.add_argument('--device', type=int, default=None, help='GPU index (default: LOCAL_RANK or 0).')


def init_logging(level):
    logging.basicConfig(level=getattr(logging, level), format='%(asctime)s %(levelname)s %(message)s', stream=sys.stderr)


def load_genome(fasta, directory, what):
    """utils.py:339-469 set_paths + :274-309 splitFasta: the genome is streamed from disk straight to
    the device by the native parser (engine.Genome.from_fasta).  As in the reference, a multi-FASTA
    given together with --adir/--bdir is also split into `<dir>/<id>.fa`; a directory alone is read
    file by file (sorted order)."""
    split_dir = None
    if fasta:
        if not os.path.isfile(fasta):
            logging.error('%s-genome fasta not found at path: %s' % (what, fasta))
            sys.exit(1)
        paths = [fasta]
        if directory:
            split_dir = os.path.abspath(directory)
            if not os.path.isdir(split_dir):
                logging.info('Creating %sdir: %s' % (what, split_dir))
                os.makedirs(split_dir, exist_ok=True)
    elif directory and os.path.isdir(directory):
        paths = [os.path.join(directory, fn) for fn in sorted(os.listdir(directory))
                 if os.path.isfile(os.path.join(directory, fn))]
    else:
        logging.error('No %s-genome fasta file provided. Quitting.' % what)
        sys.exit(1)
    try:
        G = engine.Genome.from_fasta(paths, split_dir if Dist().rank == 0 else None)
    except RuntimeError as e:
        if 'Non-unique name' in str(e):  # utils.py:300-306
            logging.error(str(e))
            sys.exit(1)
        raise
    if not G.names:
        logging.error('No sequences found for genome %s \n Cannot calculate seq lengths.' % what)
        sys.exit(1)
    return G


def start(args):
    init_logging(args.loglevel)
    dist = Dist().init()
    dev = args.device if args.device is not None else int(os.environ.get('MIMEO_FORCE_DEVICE', dist.local_rank))
    engine.init(dev)
    outdir = os.path.abspath(args.outdir) if args.outdir else os.getcwd()
    if dist.rank == 0 and not os.path.isdir(outdir):
        logging.info('Create output directory: %s' % outdir)
        os.makedirs(outdir)
    dist.barrier()
    return dist, outdir
