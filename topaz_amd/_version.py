__version__ = '0.3.18+mi355x.r1'
