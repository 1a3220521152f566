"""Test-only stand-in: the reference's optim.py imports these optimizer classes at module level; get_parameter_groups() (the
function the golden generator calls) uses none of them."""
