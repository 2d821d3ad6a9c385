"""poor man's pyflakes (no linter in the image): names loaded but never bound / imported in a file - catches the NameError a GPU-only
test would otherwise hit on the GPU box.   python tools/undefined_names.py [files...]"""
import ast
import builtins
import glob
import sys

files = sys.argv[1:] or (glob.glob("tests/*.py") + glob.glob("realise_amd/*.py") + glob.glob("tools/*.py") + ["bench.py", "__graft_entry__.py"])
bad = 0
for f in files:
    tree = ast.parse(open(f).read())
    defined = set(dir(builtins)) | {"__file__", "__name__"}
    for node in ast.walk(tree):
        if isinstance(node, (ast.Import, ast.ImportFrom)):
            for a in node.names:
                defined.add((a.asname or a.name).split(".")[0])
        elif isinstance(node, (ast.FunctionDef, ast.ClassDef)):
            defined.add(node.name)
        elif isinstance(node, ast.Name) and isinstance(node.ctx, (ast.Store, ast.Del)):
            defined.add(node.id)
        elif isinstance(node, ast.arg):
            defined.add(node.arg)
        elif isinstance(node, ast.ExceptHandler) and node.name:
            defined.add(node.name)
    for node in ast.walk(tree):
        if isinstance(node, ast.Name) and isinstance(node.ctx, ast.Load) and node.id not in defined:
            print("%s:%d: undefined name %s" % (f, node.lineno, node.id))
            bad += 1
sys.exit(1 if bad else 0)
