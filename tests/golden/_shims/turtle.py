"""Import shim (test infrastructure only): the reference's modeling/utils.py does
`from turtle import forward`, which needs tkinter. Nothing calls it."""
def forward(*a, **k):
    pass
