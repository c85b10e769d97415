"""A minimal stand-in for python-fire (not installed on either box): `run(main)` maps `--flag value`, `--flag=value`,
hyphenated or underscored flag names (the reference README uses both) onto main's keyword-only arguments, parsing
values as Python literals where possible ("True", "1e-4", "(100,50,1)", "None") and leaving the rest as strings."""
import ast
import inspect
import sys


def _value(text: str):
    try:
        return ast.literal_eval(text)
    except (ValueError, SyntaxError):
        return text


def parse(argv, fn):
    params = inspect.signature(fn).parameters
    kwargs, i = {}, 0
    while i < len(argv):
        tok = argv[i]
        if not tok.startswith("--"):
            raise SystemExit(f"unexpected argument {tok!r} (flags are --name value or --name=value)")
        name, eq, val = tok[2:].partition("=")
        name = name.replace("-", "_")
        if name not in params:
            raise SystemExit(f"unknown flag --{name}; valid flags: {', '.join(params)}")
        if eq:
            kwargs[name] = _value(val)
            i += 1
        elif i + 1 < len(argv) and not argv[i + 1].startswith("--"):
            kwargs[name] = _value(argv[i + 1])
            i += 2
        else:
            kwargs[name] = True      # bare boolean flag
            i += 1
    missing = [n for n, p in params.items() if p.default is inspect._empty and n not in kwargs]
    if missing:
        raise SystemExit(f"missing required flags: {', '.join('--' + m for m in missing)}")
    return kwargs


def run(fn, argv=None):
    return fn(**parse(sys.argv[1:] if argv is None else argv, fn))
