"""`topaz` entry point for the MI355X hot path: the sub-command registry of topaz/main.py:53-148 restricted to
the commands of the path (segment, extract, downsample, normalize, preprocess, denoise, denoise3d); `@file` argument expansion kept (main.py:55)."""
import argparse
import sys


def main(argv=None):
    from . import _version
    from .commands import denoise, denoise3d, downsample, extract, normalize, preprocess, segment
    parser = argparse.ArgumentParser(prog='topaz', formatter_class=argparse.RawDescriptionHelpFormatter,
                                     fromfile_prefix_chars='@',
                                     description='topaz on MI355X: particle extraction and denoising (inference hot path)')
    parser.add_argument('--version', action='version', version=_version.__version__)
    subparsers = parser.add_subparsers(title='commands', metavar='<command>')
    subparsers.required = True
    for module in (segment, extract, downsample, normalize, preprocess, denoise, denoise3d):
        p = subparsers.add_parser(module.name, help=module.help, fromfile_prefix_chars='@')
        module.add_arguments(p)
        p.set_defaults(func=module.main)
    args = parser.parse_args(argv)
    n = _ranks_to_launch(args)
    if n > 1:
        # `--gpus N` / `-d -2`: become the launcher -- N rank processes of this same command line, one per GPU; every
        # command shards its input list over the ranks (parallel.shard_indices), extract gathers the pick tables
        from . import parallel
        cmd = [sys.executable, '-m', 'topaz_amd'] + list(sys.argv[1:] if argv is None else argv)
        listfile, env = None, None
        if args.func is extract.main and len(args.paths) == 0:
            # the input list comes from stdin (extract.py:352 of this package; topaz/extract.py:270): N ranks sharing one stdin
            # would each consume a different part of it and take THAT for the whole list.  The launcher reads it once and
            # hands every rank the same list OUT OF BAND (a file named in TOPAZ_AMD_INPUT_LIST, read by extract_particles when
            # it has no paths): names beginning with '-' or '@', or a command line ending in a variadic option, cannot be
            # re-interpreted by the ranks' argument parsers.
            import os
            import tempfile
            names = [ln.strip() for ln in sys.stdin if ln.strip()]
            if not names:
                parser.error('no input files (none on the command line, none on stdin)')
            fd = tempfile.NamedTemporaryFile('w', prefix='topaz_inputs_', suffix='.txt', delete=False)
            fd.write('\n'.join(names) + '\n')
            fd.close()
            listfile = fd.name
            env = dict(os.environ, TOPAZ_AMD_INPUT_LIST=listfile)
        try:
            return parallel.launch_local_ranks(n, cmd, env=env)
        finally:
            if listfile:
                import os
                os.unlink(listfile)
    # like topaz/main.py:148 the command's return value (denoise returns its output paths) is NOT the exit status:
    # `raise SystemExit(<list>)` would print the list and exit 1 after a successful run
    # ... but an INT is a status: a command that reports failure by returning non-zero is not turned into a success
    rc = args.func(args)
    return rc if isinstance(rc, int) and not isinstance(rc, bool) else 0


def _ranks_to_launch(args) -> int:
    """rank processes to start for this invocation (0 / 1: run in this process).  `--gpus N` asks for N; `-d -2`
    (the reference's "all GPUs", commands/denoise3d.py:102-103,117-118) for every visible device.  Never inside a
    rank process (torchrun or our own launcher set WORLD_SIZE)."""
    from . import parallel
    if parallel.under_launcher():
        return 0
    n = int(getattr(args, 'gpus', 0) or 0)
    if n == 0 and getattr(args, 'device', 0) == -2:
        import torch
        n = torch.cuda.device_count()
    return n


if __name__ == '__main__':
    sys.exit(main())
