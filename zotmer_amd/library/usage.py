"""
A small argument parser for the `zot` usage strings (docopt, which the reference uses, is not a
dependency here).  It understands what the four hot-path commands need: short options with or
without a value (possibly repeated), positional arguments, and a trailing `<name>...` list.

    spec = Spec(options={"-m": True, "-v": False, "-M": "list"}, positionals=["<k>", "<output>"], rest="<input>")
    opts = spec.parse(argv[1:], usage_text)

Option values: True = takes one value, False = flag, "list" = takes a value and may repeat.
"""
import sys


class UsageError(SystemExit):
    pass


class Spec:
    def __init__(self, options=None, positionals=(), rest=None, rest_min=1):
        self.options = dict(options or {})
        self.positionals = list(positionals)
        self.rest = rest
        self.rest_min = rest_min

    def parse(self, args, usage):
        out = {}
        for o, kind in self.options.items():
            out[o] = [] if kind == "list" else (None if kind else False)
        pos = []
        i = 0
        only_pos = False
        while i < len(args):
            a = args[i]
            if only_pos or not a.startswith("-") or a == "-":
                pos.append(a)
            elif a == "--":
                only_pos = True
            elif a in ("-h", "--help"):
                print(usage.strip("\n"))
                raise SystemExit(0)
            else:
                name, val = a[:2], a[2:]
                if name not in self.options:
                    self._die("unknown option %s" % a, usage)
                kind = self.options[name]
                if kind is False:
                    if val:
                        self._die("option %s takes no value" % name, usage)
                    out[name] = True
                else:
                    if not val:
                        i += 1
                        if i >= len(args):
                            self._die("option %s needs a value" % name, usage)
                        val = args[i]
                    if kind == "list":
                        out[name].append(val)
                    else:
                        out[name] = val
            i += 1
        need = len(self.positionals)
        if len(pos) < need or (self.rest is None and len(pos) > need) or \
                (self.rest is not None and len(pos) - need < self.rest_min):
            self._die("wrong number of arguments", usage)
        for nm, v in zip(self.positionals, pos):
            out[nm] = v
        if self.rest is not None:
            out[self.rest] = pos[need:]
        return out

    @staticmethod
    def _die(msg, usage):
        sys.stderr.write(msg + "\n" + usage.strip("\n") + "\n")
        raise UsageError(1)
