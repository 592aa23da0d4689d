"""placeholder; replaced below"""
