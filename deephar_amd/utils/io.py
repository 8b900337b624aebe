"""Console helpers the reference's harness calls through `from deephar.utils import *` (deephar/utils/io.py,
fs.py): ANSI colour constants and four print variants.  Kept for API compatibility only."""
import os
import sys

HEADER, OKBLUE, OKGREEN, WARNING, FAIL, ENDC = '\033[95m', '\033[94m', '\033[92m', '\033[93m', '\033[91m', '\033[0m'


def printc(color, vmsg):
    sys.stdout.write(color + vmsg + ENDC)
    sys.stdout.flush()


def printcn(color, vmsg):
    printc(color, vmsg + '\n')


def printnl(vmsg):
    sys.stdout.write(vmsg + '\n')
    sys.stdout.flush()


def warning(vmsg):
    sys.stderr.write(WARNING + vmsg + ENDC + '\n')
    sys.stderr.flush()


def sprintcn(color, vmsg):
    return color + vmsg + ENDC + '\n'


def mkdir(path):
    if not os.path.isdir(path):
        os.mkdir(path)
