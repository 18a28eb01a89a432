"""
Usage:
    zot [options] <command> [<args>...]

options:
    --help          print usage information
    -V, --version   print version information
"""
# Front door, same contract as zotmer/cli.py:21-59: `zot <command> args...` imports
# commands.<command> and calls its main(argv) with argv[0] = the command name; `zot help [<command>]`
# lists or documents the commands; an unknown command is reported on stderr; Ctrl-C is swallowed.
import importlib
import pkgutil
import sys

from zotmer_amd import commands

VERSION = "Zotmer k-mer toolkit 0.1 (MI355X core)"


def available():
    return sorted(name for _, name, _ in pkgutil.iter_modules(commands.__path__))


def main_inner(argv):
    if not argv or argv[0] in ("--help", "-h"):
        print(__doc__.strip("\n"))
        return 0
    if argv[0] in ("-V", "--version"):
        print(VERSION)
        return 0
    cmd, args = argv[0], argv[1:]
    if cmd == "help" and len(args) != 1:
        print(__doc__)
        print("Available commands:")
        for name in available():
            print("\t" + name)
        print('\nuse "zot help <command>" for command specific help.')
        return 0
    if cmd == "help":
        cmd = args[0]
        try:
            print(importlib.import_module(commands.__name__ + "." + cmd).__doc__)
            return 0
        except ImportError:
            sys.stderr.write("unable to load command `%s', use `zot help` for help.\n" % cmd)
            return 1
    if cmd not in available():
        sys.stderr.write("unable to load command `%s', use `zot help` for help.\n" % cmd)
        return 1
    mod = importlib.import_module(commands.__name__ + "." + cmd)
    return mod.main([cmd] + args)


def main(argv=None):
    try:
        main_inner(sys.argv[1:] if argv is None else argv)
    except KeyboardInterrupt:
        pass
    finally:
        from zotmer_amd.library import engine
        engine.close()


if __name__ == "__main__":
    main()
