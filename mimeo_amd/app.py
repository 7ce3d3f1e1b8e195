"""`mimeo <x|self|map|filter> ...` dispatcher (reference: src/mimeo/app.py:21-63)."""
import sys
from importlib import import_module

COMMANDS = {'x': 'mimeo_amd.run_interspecies', 'self': 'mimeo_amd.run_self', 'map': 'mimeo_amd.run_map',
            'filter': 'mimeo_amd.run_filter'}


def print_usage():
    print("""
Usage: mimeo <command> [options]

Commands:
  x       Run cross-species comparison
  self    Run self-alignment analysis
  map     Run genomic mapping
  filter  Run filtering operations

For command-specific help:
  mimeo <command> --help
""")


def main():
    if len(sys.argv) < 2:
        print_usage()
        sys.exit(1)
    sub = sys.argv[1]
    if sub not in COMMANDS:
        print("Error: Unknown command '%s'" % sub)
        print_usage()
        sys.exit(1)
    sys.argv = [sys.argv[0]] + sys.argv[2:]
    try:
        import_module(COMMANDS[sub]).main()
    except ImportError as e:
        print('Error importing module %s: %s' % (COMMANDS[sub], e))
        sys.exit(1)
    except Exception as e:  # same contract as the reference: message + exit 1
        print("Error running command '%s': %s" % (sub, e))
        sys.exit(1)


if __name__ == '__main__':
    main()
